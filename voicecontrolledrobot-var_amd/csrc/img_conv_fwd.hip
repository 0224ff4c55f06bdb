// Image CNN forward of the Kuka VARPretextNet (models/pretext/arm_pretext_model.py:9-18): two fused launches,
//   img_fwd_head.hip : conv 1 + conv 2 (the first activation map stays in LDS between them)
//   img_fwd_mid.hip  : conv 3 + conv 4 + conv 5 + the image head's first Linear, one workgroup per image
// leaving act[1..5], the ReLU bits of act1 and the image head's hidden layer / partials in the workspace.
#include "var_common.h"

int launch_img_fwd(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8,
                   long bstride, const int* image_index, int B) {
    if (c->H != 84 && c->H != 96) {
        VAR_SET_ERR(c, "unsupported image size %d (84 or 96)", c->H);
        return VAR_ERR_ARG;
    }
    // 84 x 84: the role-specialised head (img_head2.hip), act1 band-tiled for img_tail2.hip; 96 x 96: round 2's kernels (NCHW act1)
    // (img_head2 walks an image's seven bands inside ONE workgroup -- the right shape for a full batch, the wrong one for the RL
    //  stage's 8 images, where the per-image latency is the kernel time: an inference-only forward of a small batch takes the
    //  round-2 head, which spreads an image's tiles over workgroups; its NCHW act1 is never read by a backward)
    const bool head2 = c->H == 84 && !(c->fwd_only && B <= 64);
    c->act1_tiled = head2;
    int rc = head2 ? launch_img_fwd_head2(c, s, params, image, is_u8, bstride, image_index, B)
                   : launch_img_fwd_head(c, s, params, image, is_u8, bstride, image_index, B);
    if (rc != VAR_OK) return rc;
    c->head_in_mid = true;
    return launch_img_fwd_mid(c, s, params, B, true);
}
