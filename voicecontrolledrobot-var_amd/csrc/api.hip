// C-ABI entry points of libvar_hip.so (see include/var_hip.h).
#include <new>
#include <stdlib.h>
#include <string.h>

#include "var_common.h"

static char g_init_err[256] = "";

#define CHECK_CTX(ctx) do { if (!(ctx)) return VAR_ERR_ARG; } while (0)
#define SET_DEVICE(ctx) VAR_HIP_CHECK(ctx, hipSetDevice((ctx)->device))

extern "C" {

int var_param_count(void) { return VAR_N_PARAMS; }

const char* var_last_error(var_ctx* ctx) { return ctx ? ctx->err : g_init_err; }

static int default_streams();

int var_init(int device_id, var_ctx** out) {
    if (!out) return VAR_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device_id < 0 || device_id >= n) {
        snprintf(g_init_err, sizeof(g_init_err), "var_init: device %d not available (%d HIP devices, %s)",
                 device_id, n, hipGetErrorString(e));
        return VAR_ERR_HIP;
    }
    var_ctx* c = new (std::nothrow) var_ctx();
    if (!c) return VAR_ERR_HIP;
    c->device = device_id;
    // Which parts of a step leave the caller's stream (bit 0: sound branch forward incl. MFCC, bit 1: sound
    // branch backward, bit 4: with bit 0, the MFCC kernel stays on the caller's stream and only the sound CNN
    // forks); var_set_streams changes the plan (0 = everything on the caller's stream, for per-kernel timing).
    c->streams = default_streams();
    c->serial = c->streams == 0;
    c->pl = make_param_layout();
    c->kl = make_pack_layout();
    if (c->pl.total != VAR_N_PARAMS) {
        snprintf(g_init_err, sizeof(g_init_err), "var_init: layout mismatch %d", c->pl.total);
        delete c;
        return VAR_ERR_ARG;
    }
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipMalloc((void**)&c->default_w.data, sizeof(float) * (size_t)c->kl.total);
    c->default_w.owner = c;
    c->bound = &c->default_w;
    c->wpack = c->default_w.data;
    if (e == hipSuccess) e = hipMalloc((void**)&c->loss_buf, sizeof(float) * 64);
    if (e == hipSuccess) e = hipMemset(c->loss_buf, 0, sizeof(float) * 64);
    c->done_ctr = (unsigned*)(c->loss_buf + 32);
    c->jsig = (unsigned*)(c->loss_buf + 40);
    if (e != hipSuccess) {
        snprintf(g_init_err, sizeof(g_init_err), "var_init: %s", hipGetErrorString(e));
        delete c;
        return VAR_ERR_HIP;
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        snprintf(g_init_err, sizeof(g_init_err), "var_init: %s", hipGetErrorString(e));
        delete c;
        return VAR_ERR_HIP;
    }
    if (mfcc_build_tables(c) != VAR_OK || pack_table_upload(c) != VAR_OK) {
        snprintf(g_init_err, sizeof(g_init_err), "var_init: %s", c->err);
        delete c;
        return VAR_ERR_HIP;
    }
    *out = c;
    return VAR_OK;
}

int var_destroy(var_ctx* c) {
    CHECK_CTX(c);
    (void)hipSetDevice(c->device);
    comm_free(c);
    mfcc_any_forget(c);
    ithor_free(c);
    armnet_free(c);
    for (auto& sl : c->ws_slot) if (sl.base) (void)hipFree(sl.base);
    for (int i = 0; i < c->n_retired; i++) (void)hipFree(c->retired[i]);
    free(c->retired);
    if (c->default_w.data) (void)hipFree(c->default_w.data);
    if (c->loss_buf) (void)hipFree(c->loss_buf);
    if (c->mfcc_tab) (void)hipFree(c->mfcc_tab);
    if (c->mfcc_psf_tab) (void)hipFree(c->mfcc_psf_tab);
    if (c->pack_segs_dev) (void)hipFree(c->pack_segs_dev);
    for (int i = 0; i < 2; i++) {
        if (c->ev_fork[i]) (void)hipEventDestroy(c->ev_fork[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->prof_ev) {
        for (int i = 0; i < 2 * kProfMaxPairs; i++) (void)hipEventDestroy(c->prof_ev[i]);
        delete[] c->prof_ev;
    }
    delete c;
    return VAR_OK;
}

int retire_block(var_ctx* c, void* p) {
    if (c->n_retired == c->cap_retired) {
        const int cap = c->cap_retired ? 2 * c->cap_retired : 8;
        void** q = (void**)realloc(c->retired, sizeof(void*) * cap);
        if (!q) { VAR_SET_ERR(c, "var_plan: out of host memory"); return VAR_ERR_HIP; }
        c->retired = q; c->cap_retired = cap;
    }
    c->retired[c->n_retired++] = p;
    return VAR_OK;
}

int var_plan_generation(var_ctx* c) { return c ? c->plan_gen : VAR_ERR_ARG; }
int var_saved_generation(var_ctx* c) { return c ? (c->saved_B > 0 ? c->saved_gen : 0) : VAR_ERR_ARG; }

int var_weights_create(var_ctx* c, var_weights** out) {
    CHECK_CTX(c);
    if (!out) return VAR_ERR_ARG;
    *out = nullptr;
    SET_DEVICE(c);
    var_weights* w = new (std::nothrow) var_weights();
    if (!w) return VAR_ERR_HIP;
    w->owner = c;
    hipError_t e = hipMalloc((void**)&w->data, sizeof(float) * (size_t)c->kl.total);
    if (e != hipSuccess) { VAR_SET_ERR(c, "var_weights_create: %s", hipGetErrorString(e)); delete w; return VAR_ERR_HIP; }
    *out = w;
    return VAR_OK;
}

int var_weights_destroy(var_ctx* c, var_weights* w) {
    CHECK_CTX(c);
    if (!w || w->owner != c || w == &c->default_w) { VAR_SET_ERR(c, "var_weights_destroy: not a handle of this context"); return VAR_ERR_ARG; }
    SET_DEVICE(c);
    if (c->bound == w) { c->bound = &c->default_w; c->wpack = c->default_w.data; }
    // like a superseded workspace, the image may still be referenced by captured graphs: retire it
    int rc = retire_block(c, w->data);
    delete w;
    return rc;
}

int var_weights_bind(var_ctx* c, var_weights* w) {
    CHECK_CTX(c);
    if (!w) w = &c->default_w;
    if (w->owner != c) { VAR_SET_ERR(c, "var_weights_bind: handle belongs to another context"); return VAR_ERR_ARG; }
    c->bound = w;
    c->wpack = w->data;
    return VAR_OK;
}

// every entry that reads the packed image checks that it was packed from the arena it is given
static int check_weights(var_ctx* c, const float* params, const char* who) {
    if (c->bound->params != params) {
        VAR_SET_ERR(c, "%s: the bound weight image was %s -- var_weights_bind + var_pack_weights first", who,
                    c->bound->params ? "packed from another parameter arena" : "never packed");
        return VAR_ERR_STATE;
    }
    return VAR_OK;
}

// Carve the workspace block `base` (or, with base == nullptr, only size it) for (B, img_hw) and point the context at it.
static size_t plan_layout(var_ctx* c, char* base, int max_batch, int img_hw) {
    const size_t B = (size_t)max_batch;
    int hs[6];
    hs[0] = img_hw;
    for (int i = 0; i < 5; i++) hs[i + 1] = conv_out(hs[i]);
    size_t off = 0;
    auto carve = [&](size_t nfloats) { size_t o = off; off += (nfloats * 4 + 255) & ~(size_t)255; return o; };
    size_t o_act[6], o_gact[6], o_sact[5], o_gsact[5];
    for (int l = 1; l <= 5; l++) {
        size_t n = B * kImgCh[l] * hs[l] * hs[l];
        // 84 x 84: act1 lives band by band in the forward head's LDS tile layout (img_head2.hip / img_tail2.hip):
        // [image][7 bands][32 channels][300 floats]; its gradient never exists in HBM (gact[1] is var_debug_buffer's scratch)
        if (l == 1 && img_hw == 84) n = B * kAct1TiledFloats;
        o_act[l] = carve(n);
        o_gact[l] = carve(n);
    }
    for (int l = 1; l <= 4; l++) {
        const size_t n = 2 * B * 32 * kSndT[l];
        o_sact[l] = carve(n);
        o_gsact[l] = carve(n);
    }
    const size_t o_hid_i = carve(B * kHid), o_hid_s = carve(2 * B * kHid);
    const size_t o_emb = carve(9 * B), o_emb_raw = carve(9 * B), o_gemb = carve(24 * B), o_hpart = carve(48 * B);
    const size_t o_ghid = carve(6 * B * kHid);
    const size_t n_img_slab = img_slab_floats(), n_snd_slab = snd_slab_floats();
    const size_t o_slab = carve(n_img_slab + n_snd_slab);
    const size_t o_mfcc = carve(2 * B * VAR_MFCC_FRAMES * VAR_MFCC_COEFFS);
    if (!base) return off;
    c->ws = base;
    c->ws_bytes = off;
    auto P = [&](size_t o) { return (float*)(c->ws + o); };
    for (int l = 1; l <= 5; l++) { c->act[l] = P(o_act[l]); c->gact[l] = P(o_gact[l]); }
    for (int l = 1; l <= 4; l++) { c->sact[l] = P(o_sact[l]); c->gsact[l] = P(o_gsact[l]); }
    c->hid_i = P(o_hid_i); c->hid_s = P(o_hid_s);
    c->head_part = P(o_hpart);
    c->emb = P(o_emb); c->emb_raw = P(o_emb_raw); c->gemb = P(o_gemb); c->ghid = P(o_ghid);
    c->slabs = P(o_slab);
    c->mfcc_buf = P(o_mfcc);
    c->slab_floats = n_img_slab + n_snd_slab;
    c->snd_slab_off = n_img_slab;
    c->maxB = max_batch;
    c->H = img_hw;
    for (int i = 0; i < 6; i++) c->hs[i] = hs[i];
    c->saved_B = 0;
    return off;
}

int var_plan(var_ctx* c, int max_batch, int img_hw) {
    CHECK_CTX(c);
    if (max_batch <= 0 || (img_hw != 84 && img_hw != 96)) {
        VAR_SET_ERR(c, "var_plan: batch %d / image size %d unsupported (size must be 84 or 96)", max_batch, img_hw);
        return VAR_ERR_ARG;
    }
    SET_DEVICE(c);
    if (c->ws && c->maxB >= max_batch && c->H == img_hw) return VAR_OK;
    // One workspace per image size, each only ever growing: two models of different image size alternating on one context
    // switch between their two blocks instead of allocating a fresh one per switch (round 2 replaced the block on every
    // change of size: hundreds of MB each time, none released before var_destroy).  A block that must GROW is retired, not
    // freed: HIP graphs captured against it (VARTrainer.capture_*, IntrinsicReward.capture) keep replaying on their own,
    // still valid, buffers.  Freed in var_destroy.
    var_ctx::WsSlot& slot = c->ws_slot[img_hw == 84 ? 0 : 1];
    if (!slot.base || slot.maxB < max_batch) {
        if (slot.base) { int rc = retire_block(c, slot.base); if (rc != VAR_OK) return rc; slot.base = nullptr; }
        if (c->ws && c->H == img_hw) c->ws = nullptr;          // (the active block was this slot's: just retired)
        const int Bn = slot.maxB > max_batch ? slot.maxB : max_batch;
        const size_t bytes = plan_layout(c, nullptr, Bn, img_hw);
        char* blk = nullptr;
        VAR_HIP_CHECK(c, hipMalloc((void**)&blk, bytes));
        VAR_HIP_CHECK(c, hipMemset(blk, 0, bytes));
        slot.base = blk; slot.maxB = Bn;
    }
    c->plan_gen++;                                              // the workspace pointers change: saved forwards are gone
    plan_layout(c, slot.base, slot.maxB, img_hw);
    return VAR_OK;
}

int var_pack_weights(var_ctx* c, void* stream, const float* params) {
    CHECK_CTX(c);
    if (!params) { VAR_SET_ERR(c, "var_pack_weights: null params"); return VAR_ERR_ARG; }
    SET_DEVICE(c);
    c->bound->params = params;
    return launch_pack_weights(c, (hipStream_t)stream, params);
}

static int check_plan(var_ctx* c, int B, int H, const char* who) {
    if (!c->ws) { VAR_SET_ERR(c, "%s: var_plan() has not been called", who); return VAR_ERR_PLAN; }
    if (B <= 0) { VAR_SET_ERR(c, "%s: batch %d", who, B); return VAR_ERR_ARG; }
    if (B > c->maxB || H != c->H) {
        VAR_SET_ERR(c, "%s: batch %d / size %d exceeds the plan (%d / %d)", who, B, H, c->maxB, c->H);
        return VAR_ERR_PLAN;
    }
    return VAR_OK;
}

// Fork the sound branch onto the side stream (it runs beside the image branch: the MFCC front-end
// is VALU work, the image convolutions are matrix-core work) and join it back before the heads.

struct AudioIn {            // optional in-step front-end: pcm != NULL => MFCC is computed here
    const int16_t* pcm = nullptr; const int* lens = nullptr; const int* clip_index = nullptr; int pcm_stride = 0;
};

static int fork_side(var_ctx* c, hipStream_t s, int i) {
    if (!(c->streams & (1 << i))) return VAR_OK;
    VAR_HIP_CHECK(c, hipEventRecord(c->ev_fork[i], s));
    VAR_HIP_CHECK(c, hipStreamWaitEvent(c->side, c->ev_fork[i], 0));
    return VAR_OK;
}
static int join_side(var_ctx* c, hipStream_t s, int i) {
    if (!(c->streams & (1 << i))) return VAR_OK;
    VAR_HIP_CHECK(c, hipEventRecord(c->ev_join[i], c->side));
    VAR_HIP_CHECK(c, hipStreamWaitEvent(s, c->ev_join[i], 0));
    return VAR_OK;
}

// up to five (dst, src, n) segments copied by ONE launch: the encoder outputs of var_arm_encoder_fwd (five
// hipMemcpyAsync calls cost more host time than the B = 8 forward's kernels)
struct CopySegs { float* dst[5]; const float* src[5]; int n[5]; int count; };
__global__ void __launch_bounds__(256) copy_out_kernel(CopySegs S) {
    for (int k = 0; k < S.count; ++k)
        for (int e = blockIdx.x * 256 + threadIdx.x; e < S.n[k]; e += gridDim.x * 256) S.dst[k][e] = S.src[k][e];
}

// gemb = [g0 | g1 | g2], n floats each; a null source contributes zeros
__global__ void __launch_bounds__(256) gemb_in_kernel(const float* __restrict__ g0, const float* __restrict__ g1, const float* __restrict__ g2,
                                                      float* __restrict__ gemb, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    gemb[e] = g0 ? g0[e] : 0.f;
    gemb[n + e] = g1 ? g1[e] : 0.f;
    gemb[2 * n + e] = g2 ? g2[e] : 0.f;
}

// dev_join: the caller (the fused training step) goes straight on into encoder_bwd, whose first kernel on `s` is the only
// consumer of the side stream's results: no join here -- that kernel waits for the sound heads on the device (heads.hip)
static int encoder_fwd(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8,
                       long bstride, const int* image_index, const float* pos, const float* neg,
                       const AudioIn* audio, int B, bool finish = true, bool dev_join = false) {
    int rc;
    if (audio && audio->pcm) {
        pos = c->mfcc_buf;
        neg = c->mfcc_buf + (size_t)B * VAR_MFCC_FRAMES * VAR_MFCC_COEFFS;
    }
    const bool snd = pos || neg;
    // small batches (the RL stage's 8 envs) stay on the caller's stream: every kernel is a few us there and a
    // fork / join pair costs more than the overlap gives
    const bool fork_ok = (c->streams & 1) && B > 32;
    hipStream_t ss = fork_ok ? c->side : s;
    // (decided before the first launch: the image forward's last kernel raises the flag the sound rows wait for)
    // Only under stream capture: in a replayed graph both branches sit in the device's queues before either starts, so a wait can
    // only be as long as the other branch's kernels; launched eagerly, a host thread that is descheduled between the two branches'
    // launches would turn into a time-out and a step with stale numbers -- and eagerly the edges' cost drowns in launch overhead.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (dev_join && hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    c->dev_join = dev_join && cap == hipStreamCaptureStatusActive && fork_ok && (c->streams & 2) && !(c->streams & 64) && image && pos && neg;
    // Launch ORDER matters under graph replay: the chain that is enqueued first after a fork keeps the hardware
    // queue of its predecessor, the other branch pays a cross-queue hand-over (5-10 us).  So the caller's stream
    // (MFCC -> image CNN -> image head) is enqueued first and the sound branch, which has slack, afterwards.
    const bool mfcc_main = (c->streams & 16) != 0 || !fork_ok;
    bool forked = false;
    if (snd && audio && audio->pcm) {
        if (!mfcc_main) { if ((rc = fork_side(c, s, 0)) != VAR_OK) return rc; forked = true; }
        if ((rc = launch_mfcc(c, mfcc_main ? s : ss, audio->pcm, audio->lens, audio->clip_index, 2 * B, audio->pcm_stride,
                              VAR_MFCC_FRAMES, c->mfcc_buf)) != VAR_OK) return rc;
    }
    if (snd && fork_ok && !forked && (rc = fork_side(c, s, 0)) != VAR_OK) return rc;     // the side stream starts after the caller's prior work
    // The one-launch image forward (conv 1-5 + head, img_fwd_all_kernel) takes 71 us against 41.6 + 36.3 for the two launches --
    // alone.  Beside the sound branch it LOSES (0.3045 vs 0.3017 ms per step, alternating runs): the step is bound by CU time, not
    // by the image chain's latency, and the sound kernels get onto the CUs at the image kernels' boundaries -- one boundary fewer
    // pushes the sound forward behind the whole image forward.  So: image-only forwards (the frozen encoder at full batch, the
    // projection of a dataset) take the fused launch, a forward with a sound branch the two launches.
    c->fuse_fwd = c->fuse_fwd_always || !snd;
    c->mid_finish = finish;
    if (image && (rc = launch_img_fwd(c, s, params, image, is_u8, bstride, image_index, B)) != VAR_OK) return rc;
    if (image && (rc = launch_heads_fwd(c, s, s, params, B, true, false, false, finish)) != VAR_OK) return rc;
    if (snd) {
        if ((rc = launch_snd_fwd(c, ss, params, pos, neg, B)) != VAR_OK) return rc;
        if ((rc = launch_heads_fwd(c, ss, ss, params, B, false, pos != nullptr, neg != nullptr, finish)) != VAR_OK) return rc;
        if (fork_ok && !c->dev_join && (rc = join_side(c, s, 0)) != VAR_OK) return rc;
    }
    c->saved_B = B;
    c->saved_gen = ++c->fwd_gen;
    c->saved_image = image;
    c->saved_u8 = is_u8;
    c->saved_bstride = bstride;
    c->saved_index = image_index;
    c->saved_pos = pos;
    c->saved_neg = neg;
    return VAR_OK;
}

int var_arm_encoder_fwd(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8,
                        long image_bstride, const float* mfcc_pos, const float* mfcc_neg, int B, int H,
                        float* image_feat, float* pos_feat, float* neg_feat, float* image_raw, float* pos_raw,
                        int save_for_bwd) {
    CHECK_CTX(c);
    if (!params) { VAR_SET_ERR(c, "var_arm_encoder_fwd: null params"); return VAR_ERR_ARG; }
    if (image && image_bstride < 3L * H * H) { VAR_SET_ERR(c, "var_arm_encoder_fwd: image stride %ld < 3*H*H", image_bstride); return VAR_ERR_ARG; }
    int rc = check_plan(c, B, H, "var_arm_encoder_fwd");
    if (rc != VAR_OK) return rc;
    if ((rc = check_weights(c, params, "var_arm_encoder_fwd")) != VAR_OK) return rc;
    SET_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    c->fwd_only = save_for_bwd == 2;                // inference with the small-batch kernels (img_conv_fwd.hip)
    // the normalised embeddings go to the caller's buffers from the heads' finish kernels themselves
    c->out_img = image ? image_feat : nullptr; c->out_pos = mfcc_pos ? pos_feat : nullptr; c->out_neg = mfcc_neg ? neg_feat : nullptr;
    rc = encoder_fwd(c, s, params, image, image_is_u8, image_bstride, nullptr, mfcc_pos, mfcc_neg, nullptr, B);
    c->fwd_only = false;
    c->out_img = c->out_pos = c->out_neg = nullptr;
    if (rc != VAR_OK) return rc;
    if (save_for_bwd != 1) c->saved_B = 0;
    CopySegs S{};
    auto seg = [&](float* dst, const float* src, int n) { S.dst[S.count] = dst; S.src[S.count] = src; S.n[S.count] = n; S.count++; };
    if (image && image_raw) seg(image_raw, c->act[5], kImgFeat * B);
    if (mfcc_pos && pos_raw) seg(pos_raw, c->sact[4], kSndFeat * B);
    if (S.count) {
        const int grid = (kImgFeat * B + 255) / 256;
        hipLaunchKernelGGL(copy_out_kernel, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, S);
        VAR_HIP_CHECK(c, hipGetLastError());
    }
    return VAR_OK;
}

__global__ void __launch_bounds__(256) row_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int rows, int dim,
                                                      float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int k = 0; k < dim; ++k) s += a[(size_t)r * dim + k] * b[(size_t)r * dim + k];
    out[r] = s;
}

int var_row_dot(var_ctx* c, void* stream, const float* a, const float* b, int rows, int dim, float* out) {
    CHECK_CTX(c);
    if (!a || !b || !out || rows < 1 || dim < 1 || dim > 64) { VAR_SET_ERR(c, "var_row_dot: bad argument (rows %d, dim %d)", rows, dim); return VAR_ERR_ARG; }
    SET_DEVICE(c);
    hipLaunchKernelGGL(row_dot_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, rows, dim, out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int var_set_reward_dot(var_ctx* c, const float* goal_feat, float* reward_out) {
    CHECK_CTX(c);
    if ((goal_feat == nullptr) != (reward_out == nullptr)) { VAR_SET_ERR(c, "var_set_reward_dot: both pointers or neither"); return VAR_ERR_ARG; }
    c->dot_with = goal_feat;
    c->dot_out = reward_out;
    return VAR_OK;
}

// fused: the head rows finish the embeddings and form the triplet gradient themselves (heads.hip); the loss value
// is computed on the side branch
static int encoder_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, bool fused = false,
                       float margin = 0.f, float inv_count = 0.f, float* loss_out = nullptr) {
    const int B = c->saved_B;
    int rc;
    if (!c->saved_image || !c->saved_pos || !c->saved_neg) {
        // a branch that did not run contributes zero gradient
        if ((rc = var_zero_async(c, s, grads, sizeof(float) * VAR_N_PARAMS)) != VAR_OK) return rc;
    }
    const int snd_lo = c->saved_pos ? 0 : B, snd_hi = c->saved_neg ? 2 * B : B;
    // streams: s = image head + the image backward chain, side = sound head + sound CNN backward + loss value
    hipStream_t ss = (c->streams & 2) ? c->side : s;
    // (same ordering rule as in the forward: the caller's chain first, then the side branch)
    // (dev_join: the side stream has not been joined since the forward's fork and needs no edge from `s` either: its rows kernel
    // waits for the image partials on the device)
    if (!c->dev_join && (rc = fork_side(c, s, 1)) != VAR_OK) return rc;
    if (c->saved_image) {
        if ((rc = launch_heads_bwd(c, s, s, params, grads, B, true, 0, 0, fused, margin, inv_count, fused ? loss_out : nullptr)) != VAR_OK) return rc;
        if ((rc = launch_img_bwd(c, s, params, grads, B)) != VAR_OK) return rc;
    }
    if ((rc = launch_heads_bwd(c, ss, ss, params, grads, B, false, snd_lo, snd_hi, fused, margin, inv_count)) != VAR_OK) return rc;
    // (fused: the loss value is summed by the image head's backward from the terms its rows kernel leaves -- heads.hip)
    if ((rc = launch_snd_bwd(c, ss, params, grads, B)) != VAR_OK) return rc;
    return join_side(c, s, 1);
}

int var_arm_encoder_bwd(var_ctx* c, void* stream, const float* params, const float* g_image_feat,
                        const float* g_pos_feat, const float* g_neg_feat, float* grads) {
    CHECK_CTX(c);
    if (!params || !grads) { VAR_SET_ERR(c, "var_arm_encoder_bwd: null params/grads"); return VAR_ERR_ARG; }
    if (c->saved_B <= 0) { VAR_SET_ERR(c, "var_arm_encoder_bwd: no forward saved (save_for_bwd=1 required)"); return VAR_ERR_STATE; }
    { int rc = check_weights(c, params, "var_arm_encoder_bwd"); if (rc != VAR_OK) return rc; }
    SET_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    const int B = c->saved_B;
    // the three embedding gradients -> gemb in ONE launch (a branch without a gradient: zeros): three copy launches were 17 us
    // of a 354-us in-batch step
    hipLaunchKernelGGL(gemb_in_kernel, dim3((3 * B + 255) / 256), dim3(256), 0, s, g_image_feat, g_pos_feat, g_neg_feat, c->gemb, 3 * B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return encoder_bwd(c, s, params, grads);
}

int var_triplet_fwd_bwd(var_ctx* c, void* stream, const float* a, const float* p, const float* n, int B,
                        float margin, float inv_count, float* loss_out, float* ga, float* gp, float* gn) {
    CHECK_CTX(c);
    if (!a || !p || !n || !loss_out || B <= 0) { VAR_SET_ERR(c, "var_triplet_fwd_bwd: bad argument"); return VAR_ERR_ARG; }
    SET_DEVICE(c);
    return launch_triplet(c, (hipStream_t)stream, a, p, n, B, margin, inv_count, loss_out, ga, gp, gn);
}

static int loss_grad_impl(var_ctx* c, hipStream_t s, const float* params, const void* image, int image_is_u8,
                          long image_bstride, const int* image_index, const float* mfcc_pos, const float* mfcc_neg,
                          const AudioIn* audio, int B, int H, float margin, float inv_count, float* grads,
                          float* loss_out, float* feats_out, const char* who) {
    if (image_bstride < 3L * H * H) { VAR_SET_ERR(c, "%s: image stride %ld < 3*H*H", who, image_bstride); return VAR_ERR_ARG; }
    int rc = check_plan(c, B, H, who);
    if (rc != VAR_OK) return rc;
    if ((rc = check_weights(c, params, who)) != VAR_OK) return rc;
    SET_DEVICE(c);
    // The training step proper (all three branches, no embedding output requested) takes the fused path: no finish
    // and no triplet kernel on the caller's chain.  (With feats_out the separate finish / triplet kernels run:
    // tests/test_gpu_round2.py::test_config2_batch256_full_batch_parity covers both.)
    const bool fused = !feats_out && image && (mfcc_pos || (audio && audio->pcm)) && (mfcc_neg || (audio && audio->pcm));
    if ((rc = encoder_fwd(c, s, params, image, image_is_u8, image_bstride, image_index, mfcc_pos, mfcc_neg, audio, B,
                          !fused, fused)) != VAR_OK) return rc;
    if (fused) {
        rc = encoder_bwd(c, s, params, grads, true, margin, inv_count, loss_out);
        c->dev_join = false;
        return rc;
    }
    if ((rc = launch_triplet(c, s, c->emb, c->emb + 3 * B, c->emb + 6 * B, B, margin, inv_count, loss_out,
                             c->gemb, c->gemb + 3 * B, c->gemb + 6 * B)) != VAR_OK) return rc;
    if (feats_out)
        if ((rc = var_copy_async(c, s, feats_out, c->emb, sizeof(float) * 9 * (size_t)B)) != VAR_OK) return rc;
    return encoder_bwd(c, s, params, grads);
}

int var_arm_loss_grad(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8,
                      long image_bstride, const float* mfcc_pos, const float* mfcc_neg, int B, int H,
                      float margin, float inv_count, float* grads, float* loss_out, float* feats_out) {
    CHECK_CTX(c);
    if (!params || !image || !mfcc_pos || !mfcc_neg || !grads || !loss_out) {
        VAR_SET_ERR(c, "var_arm_loss_grad: null argument");
        return VAR_ERR_ARG;
    }
    return loss_grad_impl(c, (hipStream_t)stream, params, image, image_is_u8, image_bstride, nullptr, mfcc_pos,
                          mfcc_neg, nullptr, B, H, margin, inv_count, grads, loss_out, feats_out, "var_arm_loss_grad");
}

int var_arm_loss_grad_gather(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8,
                             long image_bstride, const int* image_index, const float* mfcc_pos, const float* mfcc_neg,
                             int B, int H, float margin, float inv_count, float* grads, float* loss_out,
                             float* feats_out) {
    CHECK_CTX(c);
    if (!params || !image || !mfcc_pos || !mfcc_neg || !grads || !loss_out) {
        VAR_SET_ERR(c, "var_arm_loss_grad_gather: null argument");
        return VAR_ERR_ARG;
    }
    return loss_grad_impl(c, (hipStream_t)stream, params, image, image_is_u8, image_bstride, image_index, mfcc_pos,
                          mfcc_neg, nullptr, B, H, margin, inv_count, grads, loss_out, feats_out,
                          "var_arm_loss_grad_gather");
}

int var_arm_loss_grad_pcm(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8,
                          long image_bstride, const int* image_index, const int16_t* pcm, int pcm_stride,
                          const int* clip_index, const int* lens, int B, int H, float margin, float inv_count,
                          float* grads, float* loss_out, float* feats_out) {
    CHECK_CTX(c);
    if (!params || !image || !pcm || !lens || !grads || !loss_out || pcm_stride <= 0) {
        VAR_SET_ERR(c, "var_arm_loss_grad_pcm: null argument");
        return VAR_ERR_ARG;
    }
    AudioIn a;
    a.pcm = pcm; a.lens = lens; a.clip_index = clip_index; a.pcm_stride = pcm_stride;
    return loss_grad_impl(c, (hipStream_t)stream, params, image, image_is_u8, image_bstride, image_index, nullptr,
                          nullptr, &a, B, H, margin, inv_count, grads, loss_out, feats_out, "var_arm_loss_grad_pcm");
}

int var_adam_step(var_ctx* c, void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    CHECK_CTX(c);
    if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) {
        VAR_SET_ERR(c, "var_adam_step: bad argument");
        return VAR_ERR_ARG;
    }
    SET_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_adam(c, s, params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step);
    if (rc != VAR_OK) return rc;
    if (n == VAR_N_PARAMS && c->bound->params == params) return launch_pack_weights(c, s, params);
    return VAR_OK;
}

int var_adam_step_dev(var_ctx* c, void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                      long n, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      int* step_dev) {
    return var_adam_step_graph(c, stream, params, grads, exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps,
                               weight_decay, step_dev, nullptr, 0, 0, nullptr, nullptr, 0);
}

int var_adam_step_graph(var_ctx* c, void* stream, float* params, const float* grads, float* exp_avg,
                        float* exp_avg_sq, long n, const float* lr_dev, float beta1, float beta2, float eps,
                        float weight_decay, int* step_dev, const int* index_table, int row_ints, int n_rows,
                        int* cursor_dev, int* index_row, int ahead_from) {
    CHECK_CTX(c);
    if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev || n <= 0) {
        VAR_SET_ERR(c, "var_adam_step_dev: bad argument");
        return VAR_ERR_ARG;
    }
    if (index_table && (!cursor_dev || !index_row || row_ints <= 0 || n_rows <= 0)) {
        VAR_SET_ERR(c, "var_adam_step_graph: index table without cursor / row buffer / sizes");
        return VAR_ERR_ARG;
    }
    SET_DEVICE(c);
    return launch_adam_dev(c, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps,
                           weight_decay, step_dev, n == VAR_N_PARAMS && c->bound->params == params, index_table, row_ints, n_rows, cursor_dev, index_row,
                           ahead_from);
}

int var_mfcc(var_ctx* c, void* stream, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
             int pcm_stride, int out_frames, float* out) {
    CHECK_CTX(c);
    if (!pcm || !lens || !out || nclips <= 0 || pcm_stride <= 0 || out_frames <= 0) {
        VAR_SET_ERR(c, "var_mfcc: bad argument");
        return VAR_ERR_ARG;
    }
    SET_DEVICE(c);
    return launch_mfcc(c, (hipStream_t)stream, pcm, lens, clip_index, nclips, pcm_stride, out_frames, out);
}

int var_mfcc_ex(var_ctx* c, void* stream, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                int pcm_stride, int out_frames, int n_fft, int win_length, int hop_length, float* out) {
    CHECK_CTX(c);
    if (!pcm || !lens || !out || nclips <= 0 || pcm_stride <= 0 || out_frames <= 0) {
        VAR_SET_ERR(c, "var_mfcc_ex: bad argument");
        return VAR_ERR_ARG;
    }
    SET_DEVICE(c);
    if (n_fft == 512 && win_length == 400 && hop_length == 160)
        return launch_mfcc(c, (hipStream_t)stream, pcm, lens, clip_index, nclips, pcm_stride, out_frames, out);
    return launch_mfcc_any(c, (hipStream_t)stream, pcm, lens, clip_index, nclips, pcm_stride, out_frames, n_fft, win_length,
                           hop_length, out);
}

// what each profiled launch computes (both image sizes; bench.py maps (tag, image size) to the kernel's name)
static const char* kTagNames[TAG_COUNT] = {
    "(unused)", "img conv1+conv2 forward", "img conv3+4+5 forward + image head", "(unused)",
    "(unused)", "(unused)", "(unused)", "img conv3+4+5 weight gradients",
    "(unused)", "(unused)", "(unused)",
    "img conv2 data gradient + conv2, conv1 weight gradients",
    "img conv5-4-3 data gradient chain", "(unused)", "(unused)", "img weight-gradient slab fold",
    "snd_fwd_kernel", "snd_dgrad_kernel", "snd_wgrad_kernel", "snd_reduce_kernel", "heads_fwd_kernel",
    "heads_bwd_rows_kernel", "heads_bwd_gemm_kernel", "triplet_kernel", "adam_kernel", "pack_weights_kernel",
    "mfcc_kernel", "ithor conv 11x5 s2 forward", "ithor conv 11x5 s2 data gradient", "ithor conv 11x5 s2 weight gradient"};

int var_profile_tag_count(void) { return TAG_COUNT; }
const char* var_profile_tag_name(int tag) { return (tag >= 0 && tag < TAG_COUNT) ? kTagNames[tag] : ""; }

/* Record HIP events (on the launch stream) around every launch of kernel family `tag`
 * from now on (-1 = off).  Not for use under graph capture. */
static int default_streams() { return kDefaultStreams; }

int var_set_streams(var_ctx* c, int mask) {
    if (!c) return -1;
    const int old = c->streams;
    c->streams = mask < 0 ? default_streams() : (mask & (19 | 64));   // bit 6: keep the graph edge between the forward's two streams
    c->serial = c->streams == 0;
    c->fuse_fwd_always = mask >= 0 && (mask & 32);   // bit 5: the one-launch image forward even beside a sound branch (A/B timing, tests)
    return old;
}

int var_join_status(var_ctx* c, unsigned* timeouts) {
    CHECK_CTX(c);
    if (!timeouts) { VAR_SET_ERR(c, "var_join_status: null argument"); return VAR_ERR_ARG; }
    SET_DEVICE(c);
    unsigned w[8];
    VAR_HIP_CHECK(c, hipMemcpy(w, c->jsig, sizeof(w), hipMemcpyDeviceToHost));
    *timeouts = w[3] + w[7];
    return VAR_OK;
}

int var_profile_select(var_ctx* c, int tag) {
    CHECK_CTX(c);
    SET_DEVICE(c);
    if (tag >= TAG_COUNT) { VAR_SET_ERR(c, "var_profile_select: bad tag %d", tag); return VAR_ERR_ARG; }
    if (tag >= 0 && !c->prof_ev) {
        c->prof_ev = new hipEvent_t[2 * kProfMaxPairs];
        for (int i = 0; i < 2 * kProfMaxPairs; i++) VAR_HIP_CHECK(c, hipEventCreate(&c->prof_ev[i]));
    }
    c->prof_tag = tag;
    c->prof_n = 0;
    return VAR_OK;
}

/* Sum of the event-pair durations recorded since var_profile_select, and their count. Synchronises. */
int var_profile_read(var_ctx* c, float* total_ms, int* count) {
    CHECK_CTX(c);
    if (!total_ms || !count) return VAR_ERR_ARG;
    SET_DEVICE(c);
    float tot = 0.f;
    for (int i = 0; i < c->prof_n; i++) {
        float ms = 0.f;
        VAR_HIP_CHECK(c, hipEventSynchronize(c->prof_ev[2 * i + 1]));
        VAR_HIP_CHECK(c, hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *count = c->prof_n;
    return VAR_OK;
}

/* Testing hook: address and length (floats) of a workspace buffer, by name
 * ("act1".."act5", "gact1".."gact5", "sact1".."sact4", "gsact1".."gsact4", "emb", "gemb", "wpack";
 * iTHOR workspace: "ithor_s1".."ithor_s3", "ithor_gs1".."ithor_gs3": sound conv outputs / their gradients, sized for
 * the planned batch). */
int var_debug_buffer(var_ctx* c, const char* name, void** ptr, long* nfloats) {
    CHECK_CTX(c);
    if (!name || !ptr || !nfloats) return VAR_ERR_ARG;
    const size_t B = c->maxB;
    *ptr = nullptr; *nfloats = 0;
    if (!strcmp(name, "wpack")) { *ptr = c->wpack; *nfloats = c->kl.total; return VAR_OK; }
    if (!strncmp(name, "ithor_", 6)) return ithor_debug_buffer(c, name + 6, ptr, nfloats);
    if (!c->ws) { VAR_SET_ERR(c, "var_debug_buffer: no plan"); return VAR_ERR_PLAN; }
    for (int l = 1; l <= 5; l++) {
        char a[16], g[16];
        snprintf(a, sizeof a, "act%d", l); snprintf(g, sizeof g, "gact%d", l);
        const long n = (long)(B * kImgCh[l] * c->hs[l] * c->hs[l]);
        if (!strcmp(name, a)) {
            *ptr = c->act[l]; *nfloats = n;
            if (l == 1 && c->H == 84 && c->act1_tiled) {          // the tiled form -> NCHW, into the (otherwise unused) gact[1] block
                int rc = launch_act1_untile(c, nullptr, (int)B);
                if (rc != VAR_OK) return rc;
                VAR_HIP_CHECK(c, hipStreamSynchronize(nullptr));
                *ptr = c->gact[1];
            }
            return VAR_OK;
        }
        if (!strcmp(name, g)) { *ptr = c->gact[l]; *nfloats = n; return VAR_OK; }
    }
    for (int l = 1; l <= 4; l++) {
        char a[16], g[16];
        snprintf(a, sizeof a, "sact%d", l); snprintf(g, sizeof g, "gsact%d", l);
        const long n = (long)(2 * B * 32 * kSndT[l]);
        if (!strcmp(name, a)) { *ptr = c->sact[l]; *nfloats = n; return VAR_OK; }
        if (!strcmp(name, g)) { *ptr = c->gsact[l]; *nfloats = n; return VAR_OK; }
    }
    if (!strcmp(name, "hid_i")) { *ptr = c->hid_i; *nfloats = (long)B * kHid; return VAR_OK; }
    if (!strcmp(name, "hid_s")) { *ptr = c->hid_s; *nfloats = 2 * (long)B * kHid; return VAR_OK; }
    if (!strcmp(name, "emb")) { *ptr = c->emb; *nfloats = 9 * (long)B; return VAR_OK; }
    if (!strcmp(name, "mfcc")) { *ptr = c->mfcc_buf; *nfloats = 2 * (long)B * VAR_MFCC_FRAMES * VAR_MFCC_COEFFS; return VAR_OK; }
    if (!strcmp(name, "slabs")) { *ptr = c->slabs; *nfloats = (long)c->slab_floats; return VAR_OK; }
    if (!strcmp(name, "gemb")) { *ptr = c->gemb; *nfloats = 9 * (long)B; return VAR_OK; }
    VAR_SET_ERR(c, "var_debug_buffer: unknown buffer '%s'", name);
    return VAR_ERR_ARG;
}

}  // extern "C"
