// Image CNN forward of the Kuka VARPretextNet (models/pretext/arm_pretext_model.py:9-18): two fused launches,
//   img_fwd_head.hip : conv 1 + conv 2 (the first activation map stays in LDS between them)
//   img_mid3.hip     : conv 3 + conv 4 + conv 5 + the image head's first Linear, one workgroup per image
// leaving act[1..5], the ReLU bits of act1 and the image head's hidden layer / partials in the workspace.
#include "var_common.h"

namespace {
PH_DECL();      // (c3f.h's band kernel carries phase marks for `make phases`)
}
#include "c3f.h"

// conv 3..5 at 84 x 84 for a small inference-only batch (c3f.h: one (image, 16 output channels) workgroup per tile, the filter
// read in place): img_mid3 keeps an image inside ONE workgroup -- right for 256 images on 256 CUs, 34 us of per-image latency
// at the RL stage's 8
using KukaS3 = c3f::SmallCfg<32, 64, 21, 2, 11, 1, true>;
using KukaS4 = c3f::SmallCfg<64, 64, 11, 2, 6, 1, true>;
using KukaS5 = c3f::SmallCfg<64, 64, 6, 2, 3, 1, true>;
constexpr int kSmallMidB = 16;

int launch_img_fwd(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8,
                   long bstride, const int* image_index, int B) {
    if (c->H != 84 && c->H != 96) {
        VAR_SET_ERR(c, "unsupported image size %d (84 or 96)", c->H);
        return VAR_ERR_ARG;
    }
    // the role-specialised head (img_head2.hip): at 84 x 84 act1 leaves band-tiled for img_tail2.hip, at 96 x 96 as NCHW (img_tail2.hip gathers
    // its bands from the rows)
    // (img_head2 walks an image's seven bands inside ONE workgroup -- the right shape for a full batch, the wrong one for the RL
    //  stage's 8 images, where the per-image latency is the kernel time: an inference-only forward of a small batch takes the
    //  round-2 head, which spreads an image's tiles over workgroups; its NCHW act1 is never read by a backward)
    const bool head2 = !(c->fwd_only && B <= 64);
    c->act1_tiled = head2 && c->H == 84;
    if (head2 && c->H == 84 && B <= kHead2G && c->fuse_fwd) {         // one image per workgroup in both halves: the whole image forward as ONE launch
        c->head_in_mid = true;
        return launch_img_fwd_all(c, s, params, image, is_u8, bstride, image_index, B);
    }
    int rc = head2 ? launch_img_fwd_head2(c, s, params, image, is_u8, bstride, image_index, B)
                   : launch_img_fwd_head(c, s, params, image, is_u8, bstride, image_index, B);
    if (rc != VAR_OK) return rc;
    if (c->fwd_only && B <= kSmallMidB && c->H == 84) {
        const ParamLayout& L = c->pl;
        ProfScope prof(c, s, TAG_IMG_FWD0 + 2);
        if ((rc = c3f::launch_small<KukaS3>(c, s, c->act[2], params + L.img_w[2], params + L.img_b[2], c->act[3], B)) != VAR_OK) return rc;
        if ((rc = c3f::launch_small<KukaS4>(c, s, c->act[3], params + L.img_w[3], params + L.img_b[3], c->act[4], B)) != VAR_OK) return rc;
        if ((rc = c3f::launch_small<KukaS5>(c, s, c->act[4], params + L.img_w[4], params + L.img_b[4], c->act[5], B)) != VAR_OK) return rc;
        c->head_in_mid = false;                  // the image head follows as its own launch (launch_heads_fwd)
        return VAR_OK;
    }
    c->head_in_mid = true;
    return launch_img_fwd_mid(c, s, params, B, true);
}
