// Image CNN backward (autograd of models/pretext/arm_pretext_model.py:9-18 under loss.backward(), VAR/pretext_VAR.py:68): the
// order of its launches.  Per conv layer  y = relu(conv3x3_s2_p1(x, W) + b):
//   data gradients of conv 5 -> 4 -> 3: one per-image chain, img_chain.hip (both image sizes);
//   their weight gradients: ONE grid of img_wgrad.hip's split-K workgroups (below);
//   conv 2's data gradient + the weight gradients of conv 2 and conv 1: img_tail2.hip (both image sizes);
//   then one fixed-order fold of all five layers' slabs (img_wgrad.hip).
// (Round 2's kernels -- per-layer data gradients paired with the weight gradients in one grid per layer, img_bwd_tail -- served 96 x 96
//  until round 4.)
#include <stdio.h>
#include <stdlib.h>

#include "img_stage.h"
#define VAR_WGRAD_DEVICE_ONLY
#include "img_wgrad.hip"             // WgCfg, img_wgrad_body, the W84_* / W96_* configurations (device code only)
#undef VAR_WGRAD_DEVICE_ONLY

// The weight gradients of conv 5, 4 and 3 in ONE grid (84 x 84): their inputs -- gact[5..3] left by img_chain_kernel, act[4..2]
// -- are all there before it starts, and none of the three fills the GPU by itself (128 split-K workgroups each).  Longest first.
template <class W2, class W3, class W4>
__global__ void __launch_bounds__(768)
img_wgrad345_kernel(const float* __restrict__ x2, const float* __restrict__ x3, const float* __restrict__ x4,
                    const float* __restrict__ gy3, const float* __restrict__ gy4, const float* __restrict__ gy5,
                    float* __restrict__ slabs2, float* __restrict__ slabs3, float* __restrict__ slabs4, int G2, int G3, int G4, int B) {
    static_assert(W2::NW * 64 == 768 && W3::NW * 64 == 768 && W4::NW * 64 == 768, "12 waves each");
    int id = blockIdx.x;
    const int n2 = G2 * W2::NCOMBO, n3 = G3 * W3::NCOMBO;
    if (id < n2) { img_wgrad_body<W2>(x2, (long)W2::CIN * W2::H * W2::W, nullptr, gy3, slabs2, B, id % G2, id / G2, G2); return; }
    id -= n2;
    if (id < n3) { img_wgrad_body<W3>(x3, (long)W3::CIN * W3::H * W3::W, nullptr, gy4, slabs3, B, id % G3, id / G3, G3); return; }
    id -= n3;
    img_wgrad_body<W4>(x4, (long)W4::CIN * W4::H * W4::W, nullptr, gy5, slabs4, B, id % G4, id / G4, G4);
}

template <class W2, class W3, class W4>
static int launch_wgrad345(var_ctx* c, hipStream_t s, int B) {
    ProfScope prof(c, s, TAG_IMG_WGRAD0 + 2);
    constexpr int LDS_BYTES = W2::LDS_BYTES > W3::LDS_BYTES ? (W2::LDS_BYTES > W4::LDS_BYTES ? W2::LDS_BYTES : W4::LDS_BYTES)
                                                           : (W3::LDS_BYTES > W4::LDS_BYTES ? W3::LDS_BYTES : W4::LDS_BYTES);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_wgrad345_kernel<W2, W3, W4>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    auto groups = [&](int layer, int nb, int nu) {
        const int need = (B * nb + nu - 1) / nu, gmax = img_wgrad_groups(layer);
        return need < gmax ? need : gmax;
    };
    const int G2 = groups(2, W2::NB, W2::NU), G3 = groups(3, W3::NB, W3::NU), G4 = groups(4, W4::NB, W4::NU);
    c->wg_groups[2] = G2; c->wg_groups[3] = G3; c->wg_groups[4] = G4;
    hipLaunchKernelGGL((img_wgrad345_kernel<W2, W3, W4>), dim3(G2 * W2::NCOMBO + G3 * W3::NCOMBO + G4 * W4::NCOMBO), dim3(768),
                       LDS_BYTES, s, c->act[2], c->act[3], c->act[4], c->gact[3], c->gact[4], c->gact[5],
                       c->slabs + img_slab_offset(2), c->slabs + img_slab_offset(3), c->slabs + img_slab_offset(4), G2, G3, G4, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// The image backward on stream s
int launch_img_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B) {
    int rc;
    const int H = c->H;
    if (H != 84 && H != 96) { VAR_SET_ERR(c, "unsupported image size %d (84 or 96)", H); return VAR_ERR_ARG; }
    // data gradients of conv 5 -> 4 -> 3 as one per-image chain (img_chain.hip), then their three weight gradients in one grid
    if ((rc = launch_img_bwd_chain(c, s, B)) != VAR_OK) return rc;
    rc = H == 84 ? launch_wgrad345<W84_2, W84_3, W84_4>(c, s, B) : launch_wgrad345<W96_2, W96_3, W96_4>(c, s, B);
    if (rc != VAR_OK) return rc;
    // wgrad 1 + dgrad 1 + wgrad 0 in one role-specialised kernel (img_tail2.hip; act1 band-tiled at 84 x 84, NCHW at 96 x 96)
    rc = launch_img_bwd_tail2(c, s, B);
    if (rc != VAR_OK) return rc;
    (void)params;
    return launch_img_wgrad_reduce(c, s, grads, 0, 4);
}
