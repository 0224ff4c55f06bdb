// Image CNN forward: 5 x [Conv2d 3x3 stride 2 pad 1 + bias + ReLU]
// (models/pretext/arm_pretext_model.py:9-18), hand-written for gfx950.
//
// Implicit GEMM on the f32 matrix cores, D[n][pixel] = sum_k W[n][k] * X[k][pixel]:
//   A operand = filter  (rows = output channel n),  read from the packed image Wf[k][n] (L2-resident)
//   B operand = input   (cols = output pixel),      read from an LDS-staged band of input rows
//   k = tap*CIN + c, two k per v_mfma_f32_32x32x2_f32 (lane>>5 selects which).
// A workgroup owns NU "units"; a unit is R output rows (full width) of one image, whose
// 2R+1 input rows x CIN planes are staged once into LDS ([c][row][col], left/right/top/bottom
// zero padding materialised), so every input element is fetched from HBM/L2 once per unit and
// then reused by up to 9 taps x COUT channels from LDS.  Output pixels of all units are
// flattened and cut into 32-pixel blocks; (pixel block, 32-channel block) items are dealt
// round-robin to the NW waves.  The u8 -> f32 "/255" of dataset.py:67-68 is fused into
// the staging of the first layer.
#include <stdlib.h>

#include "img_stage.h"

// NWI waves share the (pixel block, channel block) items; KY x KC wave groups split the reduction
// (KY = 3: one filter row each, KC: channel chunks) and are folded through LDS in a fixed order.
template <int CIN_, int COUT_, int H_, bool U8_, int R_, int NU_, int NWI_, int KY_ = 1, int KC_ = 1>
struct FwdCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, R = R_, NU = NU_, NWI = NWI_, KY = KY_, KC = KC_;
    static constexpr int KS = KY * KC, NW = NWI * KS;
    static constexpr bool U8 = U8_;
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int IR = 2 * R + 1;            // input rows per unit (iy0 = 2*band*R - 1)
    static constexpr int PW = 2 * WO + 2;           // LDS row: col = ix + 1, cols 0..2*WO used
    static constexpr int PLANE = IR * PW;
    static constexpr int UNIT = CIN * PLANE;
    static constexpr int NB = (HO + R - 1) / R;     // bands per image
    static constexpr int PPU = R * WO;              // pixels per unit
    static constexpr int NPIX = NU * PPU;
    static constexpr int NPB = (NPIX + 31) / 32;
    static constexpr int NBLK = COUT / 32;
    static constexpr int ITEMS = NPB * NBLK;
    static constexpr int IPW = (ITEMS + NWI - 1) / NWI;
    static constexpr int RED_FLOATS = (KS - 1) * NWI * IPW * 1024;     // partial tiles of the K slices 1..KS-1
    static constexpr int LDS_FLOATS = ((NU * UNIT > RED_FLOATS ? NU * UNIT : RED_FLOATS) + 3) / 4 * 4;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int KSTEPS = (CIN * 9 + 1) / 2;
    static_assert(ITEMS % NWI == 0, "every wave must own the same number of (pixel block, channel block) items");
    static_assert(KY == 1 || KY == 3, "filter rows split 1 or 3 ways");
    static_assert(CIN < 32 || (CIN / 2) % KC == 0, "channel chunks must be whole k-steps");
    static_assert(CIN >= 32 || KS == 1, "the first layer is not K-split");
};

template <class C>
__global__ void __launch_bounds__(C::NW * 64)
img_conv_fwd_kernel(const void* __restrict__ xin, long bstride, const int* __restrict__ bidx,
                    const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
                    uint16_t* __restrict__ relu_bits, int B, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int unit0 = blockIdx.x * C::NU;

    // ---- stage the input bands: zero the pads once, then wide unrolled copies ----
    // (only the pad columns need clearing: col 0 and cols W+1.. ; every data cell, invalid rows and
    // missing units included, is written by the staging pass)
    lds_zero_cols<NT>(lds, C::NU * C::CIN * C::IR, C::PW, 0, 1, tid);
    lds_zero_cols<NT>(lds, C::NU * C::CIN * C::IR, C::PW, C::W + 1, C::PW - C::W - 1, tid);
#pragma unroll 1
    for (int u = 0; u < C::NU; ++u) {
        const int unit = unit0 + u;
        const bool uvalid = unit < total_units;
        const int bo = uvalid ? unit / C::NB : 0, band = unit % C::NB;
        const int b = bidx ? bidx[bo] : bo;            // optional batch gather (dataset row of sample bo)
        const void* img = C::U8 ? (const void*)((const uint8_t*)xin + (size_t)b * bstride)
                                : (const void*)((const float*)xin + (size_t)b * bstride);
        if (!(dbg & 2))
        stage_x_band<C::CIN, C::H, C::W, C::IR, C::PW, C::PLANE, C::U8, NT>(lds + u * C::UNIT, img,
                                                                            2 * band * C::R - 1, uvalid, tid);
    }
    __syncthreads();

    // ---- per-item lane constants ----
    const int wv = wave % C::NWI, ks = wave / C::NWI;     // item group, K slice
    int pixoff[C::IPW];
    const float* wl[C::IPW];
    f32x16 acc[C::IPW];
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
        const int it = wv + C::NWI * i;
        const int pb = it % C::NPB, nb = (it / C::NPB) % C::NBLK;
        int p = pb * 32 + l31;
        if (p >= C::NPIX) p = 0;
        const int u = p / C::PPU, q = p - u * C::PPU;
        const int oyl = q / C::WO, ox = q - oyl * C::WO;
        pixoff[i] = u * C::UNIT + (2 * oyl) * C::PW + 2 * ox;
        wl[i] = wp + nb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }

    if constexpr (C::CIN % 2 == 0) {
        // Filter values come straight from the packed image Wf[k][n] in L2; they are prefetched
        // one block (U k-steps) ahead into a second register set so that no MFMA waits on L2.
        // This wave's K slice: filter rows ky in [kyi*NKY, +NKY), channels [kci*CC, +CC).
        constexpr int NKY = 3 / C::KY;                   // filter rows per slice
        constexpr int CC = C::CIN / C::KC;               // channels per slice
        constexpr int SPT = CC / 2;                      // k-steps per tap and slice
        constexpr int U = SPT > 16 ? 16 : SPT;           // k-steps per block
        constexpr int BPT = SPT / U;                     // blocks per tap
        constexpr int NBK = NKY * 3 * BPT;
        const int kyi = ks % C::KY, kci = ks / C::KY;
        float wbuf[2][C::IPW][U];
#pragma unroll
        for (int i = 0; i < C::IPW; ++i) {
            pixoff[i] += half * C::PLANE + kyi * NKY * C::PW + kci * CC * C::PLANE;
            wl[i] += (half + (kyi * NKY * 3) * C::CIN + kci * CC) * C::COUT;
        }
#pragma unroll
        for (int i = 0; i < C::IPW; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) wbuf[0][i][u] = wl[i][(2 * u) * C::COUT];
        if (!(dbg & 1))
#pragma unroll
        for (int blk = 0; blk < NBK; ++blk) {
            const int tap = blk / BPT, c2b = (blk % BPT) * U;       // tap relative to the slice: (ky', kx)
            const int toff = (tap / 3) * C::PW + (tap % 3);
            if (blk + 1 < NBK) {
                const int ntap = (blk + 1) / BPT, nc2b = ((blk + 1) % BPT) * U;
#pragma unroll
                for (int i = 0; i < C::IPW; ++i)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        wbuf[(blk + 1) & 1][i][u] = wl[i][(ntap * C::CIN + 2 * (nc2b + u)) * C::COUT];
            }
            // keep the prefetch loads ABOVE this block's MFMAs: hipcc's scheduler otherwise sinks each
            // load down to its first use and every MFMA then waits a full L2 round trip
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < C::IPW; ++i) {
                    // no per-item predicate here: a conditional MFMA makes hipcc shuttle the whole
                    // accumulator through v_accvgpr moves around every instruction (ITEMS % NWI == 0)
                    const float bv = lds[pixoff[i] + 2 * (c2b + u) * C::PLANE + toff];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wbuf[blk & 1][i][u], bv, acc[i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (C::KS > 1) {
            // fold the K slices: slices 1.. park their tiles in LDS (the input bands are dead now),
            // slice 0 adds them in slice order and runs the epilogue
            __syncthreads();
            if (ks > 0) {
#pragma unroll
                for (int i = 0; i < C::IPW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        lds[(((ks - 1) * C::NWI + wv) * C::IPW + i) * 1024 + r * 64 + lane] = acc[i][r];
            }
            __syncthreads();
            if (ks > 0) return;
#pragma unroll
            for (int q = 1; q < C::KS; ++q)
#pragma unroll
                for (int i = 0; i < C::IPW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][r] += lds[(((q - 1) * C::NWI + wv) * C::IPW + i) * 1024 + r * 64 + lane];
        }
    } else {
        // CIN = 3: K = 27 (+1 zero row in the packed filter); k = 2*s + half, tap = k/3, c = k%3
        float wreg[C::IPW][C::KSTEPS];
#pragma unroll
        for (int i = 0; i < C::IPW; ++i)
#pragma unroll
            for (int s = 0; s < C::KSTEPS; ++s) wreg[i][s] = wl[i][(2 * s + half) * C::COUT];
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            constexpr int KMAX = C::CIN * 9 - 1;
            const int k0 = 2 * s, k1 = (2 * s + 1 > KMAX) ? KMAX : 2 * s + 1;
            const int o0 = (k0 % 3) * C::PLANE + ((k0 / 3) / 3) * C::PW + ((k0 / 3) % 3);
            const int o1 = (k1 % 3) * C::PLANE + ((k1 / 3) / 3) * C::PW + ((k1 / 3) % 3);
            const int o = half ? o1 : o0;
#pragma unroll
            for (int i = 0; i < C::IPW; ++i) {
                const float bv = lds[pixoff[i] + o];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[i][s], bv, acc[i], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias + ReLU, NCHW store.  D row = channel, D col (= lane&31) = pixel ----
#pragma unroll
    for (int i = 0; i < C::IPW; ++i) {
        const int it = wv + C::NWI * i;
        if (it >= C::ITEMS) continue;
        const int pb = it % C::NPB, nb = it / C::NPB;
        const int p = pb * 32 + l31;
        if (p >= C::NPIX) continue;
        const int u = p / C::PPU, q = p - u * C::PPU;
        const int unit = unit0 + u;
        if (unit >= total_units) continue;
        const int b = unit / C::NB, band = unit - b * C::NB;
        const int oy = band * C::R + q / C::WO;
        if (oy >= C::HO || (dbg & 4)) continue;
        float* yp = y + (size_t)b * C::COUT * C::HO * C::WO + band * C::R * C::WO + q;
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[i][r] + bias[n];
            yp[(size_t)n * C::HO * C::WO] = v > 0.f ? v : 0.f;
            bits |= v > 0.f ? (1u << r) : 0u;
        }
        // first layer: the ReLU pattern as one u16 per (pixel, accumulator half), for the fused backward
        // tail (img_bwd_tail.hip) -- 1/32 of the bytes of re-reading the activation there
        if constexpr (C::NBLK == 1) {
            if (relu_bits) relu_bits[((size_t)b * 2 + half) * C::HO * C::WO + band * C::R * C::WO + q] = (uint16_t)bits;
        }
    }
}

template <class C>
static int launch_one(var_ctx* c, hipStream_t s, const void* x, long bstride, const int* bidx, const float* wp,
                      const float* bias, float* y, int B, int layer) {
    ProfScope prof(c, s, TAG_IMG_FWD0 + layer);
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_conv_fwd_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const int units = B * C::NB;
    const int grid = (units + C::NU - 1) / C::NU;
    hipLaunchKernelGGL(img_conv_fwd_kernel<C>, dim3(grid), dim3(C::NW * 64), C::LDS_BYTES, s,
                       x, bstride, bidx, wp, bias, y, layer == 0 ? c->relu1 : nullptr, B, getenv("VAR_DBG") ? atoi(getenv("VAR_DBG")) : 0);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

//                 CIN COUT  H   U8    R  NU NW
using F84_1u = FwdCfg<3, 32, 84, true, 6, 1, 4>;
using F84_1f = FwdCfg<3, 32, 84, false, 6, 1, 4>;
using F84_2 = FwdCfg<32, 32, 42, false, 3, 2, 4>;
using F84_3 = FwdCfg<32, 64, 21, false, 11, 1, 4, 3, 1>;     // 1 image, 8 items on 4 waves x 3 ky slices -> 12 waves
using F84_4 = FwdCfg<64, 64, 11, false, 6, 1, 4, 3, 1>;    // 1 image, 4 items, 3 ky slices  -> 12 waves
using F84_5 = FwdCfg<64, 64, 6, false, 3, 3, 2, 3, 2>;     // 3 images, 2 items, 3 ky x 2 channel halves -> 12 waves
using F96_1u = FwdCfg<3, 32, 96, true, 4, 1, 3>;
using F96_1f = FwdCfg<3, 32, 96, false, 4, 1, 3>;
using F96_2 = FwdCfg<32, 32, 48, false, 4, 1, 3>;
using F96_3 = FwdCfg<32, 64, 24, false, 4, 2, 3>;
using F96_4 = FwdCfg<64, 64, 12, false, 6, 1, 4, 3, 1>;
using F96_5 = FwdCfg<64, 64, 6, false, 3, 3, 2, 3, 2>;

int launch_img_fwd(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8,
                   long bstride, const int* image_index, int B) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    const float* w[5];
    const float* b[5];
    for (int i = 0; i < 5; i++) { w[i] = c->wpack + K.img_f[i]; b[i] = params + L.img_b[i]; }
    int rc;
#define RUN(CFG, X, BS, I, Y) do { if ((rc = launch_one<CFG>(c, s, X, BS, (I) == 0 ? image_index : nullptr, w[I], b[I], Y, B, I)) != VAR_OK) return rc; } while (0)
    // VAR_NO_HEAD=1 (tuning aid): conv 1 and conv 2 as separate kernels through act1 in HBM
    static const bool fused_head = !getenv("VAR_NO_HEAD");
    if (c->H != 84 && c->H != 96) {
        VAR_SET_ERR(c, "unsupported image size %d (84 or 96)", c->H);
        return VAR_ERR_ARG;
    }
    if (fused_head) {
        if ((rc = launch_img_fwd_head(c, s, params, image, is_u8, bstride, image_index, B)) != VAR_OK) return rc;
    } else if (c->H == 84) {
        if (is_u8) { RUN(F84_1u, image, bstride, 0, c->act[1]); } else { RUN(F84_1f, image, bstride, 0, c->act[1]); }
        if (!getenv("VAR_NO_PIPE")) { if ((rc = launch_img_fwd_conv2_pipe(c, s, c->act[1], w[1], b[1], c->act[2], B)) != VAR_OK) return rc; }
        else RUN(F84_2, c->act[1], 32L * 42 * 42, 1, c->act[2]);
    } else {
        if (is_u8) { RUN(F96_1u, image, bstride, 0, c->act[1]); } else { RUN(F96_1f, image, bstride, 0, c->act[1]); }
        if (!getenv("VAR_NO_PIPE")) { if ((rc = launch_img_fwd_conv2_pipe(c, s, c->act[1], w[1], b[1], c->act[2], B)) != VAR_OK) return rc; }
        else RUN(F96_2, c->act[1], 32L * 48 * 48, 1, c->act[2]);
    }
    c->head_in_mid = false;
    // VAR_NO_MID=1 (tuning aid): conv 3, 4, 5 as separate kernels
    static const bool fused_mid = !getenv("VAR_NO_MID");
    if (fused_mid) {
        static const bool head_in_mid = !getenv("VAR_NO_MID_HEAD");
        c->head_in_mid = head_in_mid;
        if ((rc = launch_img_fwd_mid(c, s, params, B, head_in_mid)) != VAR_OK) return rc;
    } else if (c->H == 84) {
        RUN(F84_3, c->act[2], 32L * 21 * 21, 2, c->act[3]);
        RUN(F84_4, c->act[3], 64L * 11 * 11, 3, c->act[4]);
        RUN(F84_5, c->act[4], 64L * 6 * 6, 4, c->act[5]);
    } else {
        RUN(F96_3, c->act[2], 32L * 24 * 24, 2, c->act[3]);
        RUN(F96_4, c->act[3], 64L * 12 * 12, 3, c->act[4]);
        RUN(F96_5, c->act[4], 64L * 6 * 6, 4, c->act[5]);
    }
#undef RUN
    return VAR_OK;
}
