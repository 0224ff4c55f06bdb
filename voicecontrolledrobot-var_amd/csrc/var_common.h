// Internal definitions shared by the gfx950 kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/var_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Kuka VARPretextNet geometry (models/pretext/arm_pretext_model.py:9-56)
static constexpr int kImgCh[6] = {3, 32, 32, 64, 64, 64};
static constexpr int kHid = 128;
static constexpr int kEmb = 3;
static constexpr int kImgFeat = 576;   // 64*3*3
static constexpr int kSndFeat = 160;   // 32*5
static constexpr int kSndT[5] = {100, 48, 23, 11, 5};

// ---- parameter arena offsets (floats), state_dict() registration order ----
struct ParamLayout {
    int img_w[5], img_b[5];
    int snd_w[4], snd_b[4];
    int ih_w0, ih_b0, ih_w1, ih_b1;
    int sh_w0, sh_b0, sh_w1, sh_b1;
    int total;
};

inline ParamLayout make_param_layout() {
    ParamLayout L{};
    int o = 0;
    for (int i = 0; i < 5; i++) {
        L.img_w[i] = o; o += kImgCh[i + 1] * kImgCh[i] * 9;
        L.img_b[i] = o; o += kImgCh[i + 1];
    }
    const int snd_k[4] = {200, 96, 96, 96};
    for (int i = 0; i < 4; i++) {
        L.snd_w[i] = o; o += 32 * snd_k[i];
        L.snd_b[i] = o; o += 32;
    }
    L.ih_w0 = o; o += kHid * kImgFeat; L.ih_b0 = o; o += kHid;
    L.ih_w1 = o; o += kEmb * kHid;     L.ih_b1 = o; o += kEmb;
    L.sh_w0 = o; o += kHid * kSndFeat; L.sh_b0 = o; o += kHid;
    L.sh_w1 = o; o += kEmb * kHid;     L.sh_b1 = o; o += kEmb;
    L.total = o;
    return L;
}

// ---- packed (kernel-side) weight images, refreshed by var_pack_weights ----
// fwd image conv l : conv 1, 2: Wf[k][n], k = tap*CIN + c (conv1: K 27 padded to 28 with a zero row); conv 3..5: 1-KiB MFMA
//                    A-fragment pieces [group of 16 k][16-channel tile][lane][4] for img_mid3.hip (pack_adam.hip, types 4 / 5)
// dX  image conv l : Wd[tap][n][c]             (layers 1..4 = second..fifth conv)
// sound conv l     : Ws[k][n], k = kt*CINw + c  (conv0: k = kt*40 + f), and WsT[kt][n][c] for dX
struct PackLayout {
    int img_f[5];
    int img_d[5];     // [1] only: conv 2's transposed filter for the tail kernels ([2..4]: img_a below)
    int img_a[5];     // [0], [1] unused: Wd of layers 2..4 in MFMA A-fragment order (img_chain.hip), see below
    int snd_f[4];
    int snd_d[4];     // [0] unused
    int ih_w0t;       // image head Linear(576,128) weight transposed: [k][j]
    int sh_w0t;       // sound head Linear(160,128) weight transposed: [k][j]
    int total;
};

inline PackLayout make_pack_layout() {
    PackLayout P{};
    int o = 0;
    for (int i = 0; i < 5; i++) {
        int K = kImgCh[i] * 9; if (K & 1) K++;
        P.img_f[i] = o; o += K * kImgCh[i + 1];
    }
    P.img_d[0] = P.img_d[2] = P.img_d[3] = P.img_d[4] = -1;
    P.img_d[1] = o; o += 9 * kImgCh[2] * kImgCh[1];
    const int snd_k[4] = {200, 96, 96, 96};
    for (int i = 0; i < 4; i++) { P.snd_f[i] = o; o += snd_k[i] * 32; }
    P.snd_d[0] = -1;
    for (int i = 1; i < 4; i++) { P.snd_d[i] = o; o += 96 * 32; }
    // A-fragment order of the transposed filter, for v_mfma_f32_16x16x4_f32 with D[c][pixel]: [tap][c tile][group of 4 k-steps]
    // [lane = 16 q + l15][j]: W[tap][n = 16 sg + 4 j + q][c = 16 ct + l15] -- one 16-byte load per lane = its A operands of 4 k-steps
    P.img_a[0] = P.img_a[1] = -1;
    for (int i = 2; i < 5; i++) { P.img_a[i] = o; o += 9 * kImgCh[i + 1] * kImgCh[i]; }
    P.ih_w0t = o; o += kImgFeat * kHid;
    P.sh_w0t = o; o += kSndFeat * kHid;
    P.total = o;
    return P;
}

// profiling tags: one per kernel family (var_profile_select / var_profile_read)
enum VarTag {
    TAG_IMG_FWD0 = 0,    // ..4  img_conv_fwd_kernel, layer l
    TAG_IMG_WGRAD0 = 5,  // ..9  img_wgrad_kernel, layer l
    TAG_IMG_DGRAD0 = 10, // 11..14 img_dgrad_kernel, layer l (1..4)
    TAG_IMG_WREDUCE = 15,
    TAG_SND_FWD = 16, TAG_SND_DGRAD = 17, TAG_SND_WGRAD = 18, TAG_SND_REDUCE = 19,
    TAG_HEADS_FWD = 20, TAG_HEADS_BWD_ROWS = 21, TAG_HEADS_BWD_W = 22, TAG_TRIPLET = 23,
    TAG_ADAM = 24, TAG_PACK = 25, TAG_MFCC = 26,
    TAG_ITHOR_S2_FWD = 27, TAG_ITHOR_S2_DGRAD = 28, TAG_ITHOR_S2_WGRAD = 29,   // the 11x5 sound convolution of the iTHOR model
    TAG_COUNT = 30
};
constexpr int kProfMaxPairs = 4096;

// One packed weight image per MODEL (var_weights_create / var_weights_bind): a training model and a frozen
// copy of the encoder can share a device context without trampling each other's kernel-side filters.
struct var_weights {
    float* data = nullptr;          // PackLayout::total floats
    const float* params = nullptr;  // the arena the image was last packed from (guards against a stale binding)
    var_ctx* owner = nullptr;
};

struct var_ctx {
    int device = 0;
    char err[512] = {0};
    ParamLayout pl;
    PackLayout kl;
    // plan
    int maxB = 0, H = 0;
    int hs[6] = {0};
    char* ws = nullptr;           // the ACTIVE workspace block (= ws_slot[..].base of the planned image size), carved below
    struct WsSlot { char* base = nullptr; int maxB = 0; } ws_slot[2];   // one block per image size (84, 96), each only grows
    size_t ws_bytes = 0;
    float* wpack = nullptr;       // = bound->data: the packed image the launchers read (var_weights_bind)
    var_weights default_w;        // the context's own image, bound until the host binds a per-model one
    var_weights* bound = nullptr;
    void** retired = nullptr; int n_retired = 0, cap_retired = 0;   // superseded workspaces: kept alive until var_destroy,
                                                                    // captured graphs keep their (old, self-consistent) pointers
    int plan_gen = 0;             // bumped by every re-plan
    int fwd_gen = 0, saved_gen = 0;   // generation id of the forward whose activations the workspace holds (0 = none)
    float* act[6] = {nullptr};    // act[l] = output of image conv l (l = 1..5), post-ReLU, NCHW
    float* gact[6] = {nullptr};   // d(loss)/d(pre-activation of conv l output)
    float* sact[5] = {nullptr};   // sound activations for 2B clips: [l] l = 1..4, layout (clip, 32, T_l)
    float* gsact[5] = {nullptr};
    float* hid_i = nullptr;       // (B,128) image head hidden, post-ReLU
    float* hid_s = nullptr;       // (2B,128)
    float* emb = nullptr;         // (3B,3) normalised [img | pos | neg]
    float* emb_raw = nullptr;     // (3B,3) pre-normalise
    float* head_part = nullptr;   // (3B,4,4) partials of the 128 -> 3 layer per block of 32 hidden units
    float* gemb = nullptr;        // (3B,3) grads wrt normalised embeddings
    float* ghid = nullptr;        // (3B,128)
    void* pack_segs_dev = nullptr; int pack_nseg = 0;   // pack segment table in device memory (pack_adam.hip)
    unsigned* done_ctr = nullptr; // self-resetting block counter of the graph-replayed Adam kernel
    // Device-side hand-over between the two streams of a training step (heads.hip, join_signal below): [0] arrivals of the sound
    // heads' forward workgroups, [1] completed launches of that kernel, [2] how many of them the caller's stream has waited for,
    // [3] waits for it that timed out (sticky; var_join_status); [4..7] the same for the conv 3-5 kernel and the side stream
    unsigned* jsig = nullptr;
    const float* dot_with = nullptr;   // var_set_reward_dot: the next small-batch image-head forward also leaves <emb, dot_with> rows in dot_out
    float* dot_out = nullptr;
    bool dev_join = false;        // this step's forward left the side stream un-joined: the image rows wait on jsig[1]
    float* slabs = nullptr;       // split-K partial weight gradients
    size_t slab_floats = 0;
    size_t snd_slab_off = 0;
    int wg_groups[5] = {0};       // split-K workgroups used by the last image wgrad launches      // sound slabs start here inside `slabs`
    float* loss_buf = nullptr;
    float* mfcc_tab = nullptr;    // window / twiddles / mel / DCT tables
    // profiling: HIP events around the launches of ONE selected kernel family
    int prof_tag = -1;
    int prof_n = 0;               // pairs recorded since select
    hipEvent_t* prof_ev = nullptr;
    // saved forward
    // var_arm_encoder_fwd's embedding outputs: the finish kernels of the heads write them directly (no copy launch afterwards)
    float *out_img = nullptr, *out_pos = nullptr, *out_neg = nullptr;
    bool head_in_mid = false;             // the last image forward also ran the image head (img_mid3.hip)
    bool mid_finish = false;              // ... and is to finish the image embeddings too (encoder_fwd: finish wanted)
    bool fuse_fwd_always = false;         // var_set_streams bit 5
    bool fuse_fwd = false;                // 84 x 84, B <= 256: conv 1-5 + image head as ONE launch (img_fwd_all_kernel).  Set per call by
                                          // encoder_fwd (api.hip): image-only forwards, or always with var_set_streams bit 5 -- see there
    int saved_B = 0;
    const void* saved_image = nullptr;
    int saved_u8 = 0;
    long saved_bstride = 0;
    const float* saved_pos = nullptr;
    const float* saved_neg = nullptr;
    const int* saved_index = nullptr;     // optional image gather index of the saved forward
    // side stream for the sound branch (runs beside the image branch) and its fork/join events
    bool act1_tiled = false;              // the last image forward left act1 band-tiled (img_head2.hip) rather than NCHW
    bool fwd_only = false;                // the running forward saves nothing for a backward (var_arm_encoder_fwd, save_for_bwd = 0)
    bool serial = false;                  // var_set_streams(0): everything on the caller's stream (per-kernel profiling)
    int streams = 0;              // bit mask, see var_init
    hipStream_t side = nullptr;
    hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
    float* mfcc_buf = nullptr;            // (2*maxB, 100, 40) when the front-end runs inside the step
    float* mfcc_psf_tab = nullptr;        // tables of the python_speech_features front-end (mfcc_psf.hip), built on first use
    void* comm = nullptr; int comm_rank = 0, comm_size = 1;   // RCCL communicator (comm.hip), created by var_comm_init
    void* arm = nullptr;                  // actor-critic state (armnet.hip), created by var_armnet_plan
    void* ith = nullptr;                  // iTHOR model state (ithor.hip), created by var_ithor_plan
    const unsigned* adam_guard = nullptr; // device word: non-zero = the gradient of this step is invalid (a persistent GRU launch timed
    long adam_guard_n = 0;                // out): Adam launches over adam_guard_n parameters leave parameters, moments and step alone
    const float* adam_guard_loss = nullptr;   // device float (the step's -- under data parallelism the all-reduced -- loss): not finite = the same
                                              // (var_ithor_guard_loss; a rank's time-out reaches every rank through the NaN it sums in)
};

// MaxDynamicSharedMemorySize and friends are attributes of a function ON A DEVICE: the launchers' set-once flags are masks
static inline unsigned var_dev_bit(const var_ctx* c) { return 1u << (c->device & 31); }

#define VAR_SET_ERR(ctx, ...) do { if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); } while (0)

#define VAR_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            VAR_SET_ERR(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return VAR_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

struct ProfScope {
    var_ctx* c; hipStream_t s; bool on;
    ProfScope(var_ctx* c_, hipStream_t s_, int tag) : c(c_), s(s_), on(false) {
        if (c->prof_tag == tag && c->prof_ev && c->prof_n < kProfMaxPairs) {
            on = hipEventRecord(c->prof_ev[2 * c->prof_n], s) == hipSuccess;
        }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(c->prof_ev[2 * c->prof_n + 1], s); c->prof_n++; }
    }
};

// Phase timing inside a kernel (tuning builds only: -DVAR_PHASES): thread 0 of one workgroup accumulates the
// shader-clock cycles between PH(i) marks into g_phase[i]; var_debug_phases() reads and clears them.
#ifndef VAR_PH_THREAD
#define VAR_PH_THREAD 0
#endif
#ifdef VAR_PHASES
#define PH_DECL() static __device__ unsigned long long g_phase[32]
#define PH_INIT2(blk, thr) unsigned long long ph_t = clock64(); const bool ph_on = (int)blockIdx.x == (blk) && (int)threadIdx.x == (thr)
#define PH_INIT(blk) PH_INIT2(blk, 0)
#define PH(i) do { if (ph_on) { const unsigned long long t_ = clock64(); g_phase[i] += t_ - ph_t; ph_t = t_; } } while (0)
// register-accumulated form (PHR_INIT / PHR / PHR_FLUSH): a mark touches no memory, so it does not wait for the kernel's own
// outstanding loads and stores the way PH()'s read-modify-write of g_phase does; 16 phases, written out once at the end
// g_phase[16] / [17]: the marked wave's whole span in s_memrealtime ticks (100 MHz) and in shader cycles (s_memtime): their ratio is
// the clock the kernel ran at (MI355X_MICROARCH.md, DVFS give-back item 6; tools/phases.py prints it)
#define PHR_INIT(blk, thr) unsigned long long ph_t = clock64(), ph_a[16] = {0}; const unsigned long long ph_c0 = ph_t, ph_r0 = wall_clock64(); const bool ph_on = (int)blockIdx.x == (blk) && (int)threadIdx.x == (thr)
#define PHR(i) do { const unsigned long long t_ = clock64(); ph_a[i] += t_ - ph_t; ph_t = t_; } while (0)
#define PHR_FLUSH() do { if (ph_on) { for (int i_ = 0; i_ < 16; ++i_) g_phase[i_] += ph_a[i_]; g_phase[16] += wall_clock64() - ph_r0; g_phase[17] += clock64() - ph_c0; } } while (0)
#else
#define PH_DECL()
#define PH_INIT(blk)
#define PH_INIT2(blk, thr)
#define PH(i)
#define PHR_INIT(blk, thr)
#define PHR(i)
#define PHR_FLUSH()
#endif


// ------------------------------------------------------------------------------------------
// Device-side hand-over between the two streams of a training step (heads.hip explains when and why; sig = var_ctx::jsig or
// jsig + 4: [0] arrivals of the producer's workgroups, [1] the flag: completed producer launches, [2] how many of them the other
// stream has waited for, [3] its timed-out waits).
typedef __attribute__((address_space(1))) unsigned gu32h;
// No fences (a release fence at agent scope writes the XCD's whole L2 back, an acquire drops it -- beside the conv kernels that
// cost what the edge did): the handed-over floats themselves travel as agent-scope atomic stores and loads, which go past the
// non-coherent cache levels; the producer's barrier waits for its stores' acknowledgements (vmcnt) before one thread counts the
// workgroup in, and the consumer issues its loads only after it has seen the flag.
__device__ __forceinline__ void join_store(float* p, float v) {
    __hip_atomic_store((gu32h*)p, __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float join_load(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load((gu32h*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// The producer's side: every thread's join_store()s are issued.  The LAST workgroup to arrive counts this stream's flag up and
// then waits (one thread; 5 ms at most, counted in other[3]) until the OTHER stream's flag has been counted up once more than
// this side has consumed it (other[2]): the kernel behind this one needs that stream's results, and the wait belongs here, not
// there -- a kernel that starts by polling may be resident on every CU while what it waits for cannot be placed (heads.hip).
// Each side counts its own flag up before it waits for the other's: no cycle.  Flags only count up: nothing to lower, no
// arrival counting in the consumers (768 same-address atomics took longer than the kernels they were in).
__device__ __forceinline__ void join_signal(unsigned* sig, unsigned n_wg, unsigned* other) {
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(sig, 1u) == n_wg - 1) {
        atomicExch(sig, 0u);
        atomicAdd(sig + 1, 1u);
        const unsigned want = __hip_atomic_load((gu32h*)(other + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        const unsigned long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load((gu32h*)(other + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            if (wall_clock64() - t0 > 500000ull) { atomicAdd(other + 3, 1u); break; }     // 5 ms of the 100 MHz counter
            __builtin_amdgcn_s_sleep(8);
        }
        __hip_atomic_store((gu32h*)(other + 2), want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// ------------------------------------------------------------------------------------------

static inline int conv_out(int h) { return (h - 1) / 2 + 1; }   // 3x3 s2 p1

extern "C" int retire_block(var_ctx* c, void* p);   // api.hip: keep a superseded device block alive until var_destroy
int mfcc_build_tables(var_ctx* c);
int launch_mfcc_any(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                    int pcm_stride, int out_frames, int n_fft, int win, int hop, float* out);
int launch_mfcc_psf(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                    int pcm_stride, int out_frames, float* out);     // mfcc.hip, PSF flavour
void mfcc_any_forget(var_ctx* c);
void ithor_free(var_ctx* c);
void armnet_free(var_ctx* c);
void comm_free(var_ctx* c);
int pack_table_upload(var_ctx* c);
size_t img_slab_floats();
int img_wgrad_groups(int layer);
size_t img_slab_offset(int layer);
int launch_img_wgrad_reduce(var_ctx* c, hipStream_t s, float* grads, int lo, int hi);
size_t snd_slab_floats();

// kernels' host launchers (one per .hip file) ------------------------------------------
int launch_pack_weights(var_ctx* c, hipStream_t s, const float* params);
int launch_img_fwd(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8,
                   long bstride, const int* image_index, int B);
int launch_img_fwd_head(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8, long bstride,
                        const int* image_index, int B);
// Zero / copy device memory with a KERNEL instead of hipMemsetAsync / hipMemcpyAsync.  Measured on ROCm 7.2 (MI355X): a memset
// node of a captured graph takes its fill pattern from runtime-owned staging memory that is recycled at the next device or
// stream synchronise -- the replays after that fill with whatever is there (the iTHOR step's initial GRU state then read
// 16-byte patterns of host pointers: loss == margin, zero gradients; tools/graph_replay_check.py).  Everything the library
// enqueues may end up in a caller's graph, so it enqueues kernels only.  bytes % 4 == 0, 4-byte aligned.
int var_zero_async(var_ctx* c, hipStream_t s, void* p, size_t bytes);
int var_copy_async(var_ctx* c, hipStream_t s, void* dst, const void* src, size_t bytes);
int launch_img_fwd_head2(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8, long bstride,
                         const int* image_index, int B);      // img_head2.hip: the 84 x 84 form, two small workgroups per CU
#ifndef VAR_HEAD2_G
#define VAR_HEAD2_G 256
#endif
static constexpr int kHead2G = VAR_HEAD2_G;   // persistent workgroups of img_head2_kernel (one image = NB tiles each)
int launch_img_fwd_mid(var_ctx* c, hipStream_t s, const float* params, int B, bool with_head);
int launch_img_fwd_all(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8, long bstride,
                       const int* image_index, int B);      // img_mid3.hip: conv 1-5 + image head in one launch (84 x 84, B <= 256)
// default: the whole sound branch -- MFCC front-end, sound CNN and sound head, forward and backward -- beside the image
// CNN on one side stream.  (Round 1 kept the 61-us MFCC on the caller's stream, mask 19; with round 2's 43-us kernel the
// image chain starting at once and the front-end on the side stream is 4-5 us per step faster: 0.346 vs 0.351 ms.)  Measured on MI355X (graph replay):
// every cross-stream edge costs several us, and two GPU-filling persistent kernels side by side slow each other
// down more than the overlap gains -- only the small sound kernels are worth forking.
static constexpr int kDefaultStreams = 3;
static constexpr int kTailG = 256;    // workgroups (= layer-0 slabs) of the fused backward tail
static constexpr long kAct1TiledFloats = 7L * 32 * 300 + 128;   // act1 of one 84 x 84 image, band-tiled (+ slack for whole-KiB reads)
int launch_act1_untile(var_ctx* c, hipStream_t s, int B);      // img_head2.hip: tiled act1 -> NCHW in gact[1] (var_debug_buffer)
int launch_img_bwd_chain(var_ctx* c, hipStream_t s, int B);   // img_chain.hip: dgrad 4 -> 3 -> 2 (conv 5, 4, 3) per image, gradients resident in LDS
static constexpr int kTail2G = 256;   // persistent workgroups (= layer-0 and layer-1 slabs) of img_tail2_kernel: one image each
int launch_img_bwd_tail2(var_ctx* c, hipStream_t s, int B);   // img_tail2.hip: wgrad 2 + dgrad 2 + wgrad 1 at 84 x 84
int launch_img_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B);
int launch_snd_fwd(var_ctx* c, hipStream_t s, const float* params, const float* pos, const float* neg, int B);
int launch_snd_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B);
int launch_heads_fwd(var_ctx* c, hipStream_t s, hipStream_t ss, const float* params, int B, bool has_img, bool has_pos,
                     bool has_neg, bool finish = true);
int launch_heads_bwd(var_ctx* c, hipStream_t s, hipStream_t ss, const float* params, float* grads, int B, bool has_img,
                     int snd_lo, int snd_hi, bool fused = false, float margin = 0.f, float inv_count = 0.f, float* loss_out = nullptr);
int launch_triplet_loss(var_ctx* c, hipStream_t s, const float* params, int B, float margin, float inv_count,
                        float* loss_out);
int launch_triplet(var_ctx* c, hipStream_t s, const float* a, const float* p, const float* n, int B,
                   float margin, float inv_count, float* loss_out, float* ga, float* gp, float* gn);
int launch_adam(var_ctx* c, hipStream_t s, float* p, const float* g, float* m, float* v, long n,
                float lr, float b1, float b2, float eps, float wd, int step);
int launch_adam_dev(var_ctx* c, hipStream_t s, float* p, const float* g, float* m, float* v, long n,
                    const float* lr_dev, float b1, float b2, float eps, float wd, int* step_dev, bool repack,
                    const int* idx_table, int row_ints, int n_rows, int* cursor, int* idx_row, int ahead_from);
int launch_mfcc(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                int pcm_stride, int out_frames, float* out);

// snd_bf16.hip: the iTHOR model's 11x5 sound convolution in its bf16 mode (patch staged in LDS, 32x32x16 bf16 MFMA)
long snd_bf16_workspace_bytes(int nclips);
int snd2_bf16_fwd(var_ctx* c, hipStream_t s, const float* x, const float* w, const float* bias, float* y, int nclips,
                  int maxclips, void* ws);
int snd1_bf16_fwd(var_ctx* c, hipStream_t s, const float* x0, int n0, const float* x1, int n1, const float* w, const float* bias,
                  float* y, int maxclips, void* ws);
int snd1_bf16_wgrad(var_ctx* c, hipStream_t s, const float* x0, int n0, const float* x1, int n1, float* dw, float* slab, int maxclips,
                    void* ws);
int snd3_bf16_fwd(var_ctx* c, hipStream_t s, const float* w, const float* bias, float* y, int nclips, int maxclips, void* ws);
int snd3_bf16_dgrad(var_ctx* c, hipStream_t s, const float* gy_seq, const float* w, float* dx, float* bias_part, int* nparts,
                    int nclips, int maxclips, void* ws);
int snd2_bf16_prepare_gy(var_ctx* c, hipStream_t s, const float* gy, int nclips, int maxclips, void* ws);
int snd2_bf16_dgrad(var_ctx* c, hipStream_t s, const float* w, float* dx, float* bias_part, int* nparts, int nclips, int maxclips,
                    void* ws);
int snd2_bf16_wgrad(var_ctx* c, hipStream_t s, float* dw, float* slab, int nclips, int maxclips, void* ws);
int snd3_bf16_wgrad(var_ctx* c, hipStream_t s, float* dw, float* slab, int nclips, int maxclips, void* ws);   // after snd3_bf16_dgrad
int ithor_debug_buffer(var_ctx* c, const char* name, void** ptr, long* nfloats);

// gru_bf16.hip: one launch per GRU time step and pass (recurrent product on bf16 MFMA + gate arithmetic), bf16 mode
long gru_bf16_workspace_bytes(int maxclips);
void* gru_bf16_h16(void* ws);                 // bf16 copies: states (dir, step 0..73, clip, 512)
void* gru_bf16_dgh16(void* ws, int maxclips); //   gate gradients wrt gh (dir, step, clip, 1536)
void* gru_bf16_dgi16(void* ws, int maxclips); //   gate gradients wrt gi (dir, clip*73+t, 1536)
void* gru_bf16_x16(void* ws, int maxclips);   //   the input sequence (clip*73+t, 448)
int gru_bf16_to_bf16(var_ctx* c, hipStream_t s, const float* x, void* y, long n);     // n % 8 == 0
int gru_bf16_convert_x(var_ctx* c, hipStream_t s, const float* x, long n, int maxclips, void* ws);
void* gru_bf16_wih16(void* ws, int maxclips); //   W_ih (dir, 1536, 448)
// fp32 (dir, clip slice of 64, [b_ih: r z n | b_hh: r z n], 512): per-slice bias-gradient sums left by gru_bf16_seq_bwd
float* gru_bf16_bias_part(void* ws, int maxclips);
// once per forward, before the input projection, ONE launch: W_hh's fragment tables, W_ih's bf16 copy, the zero initial
// state (row 0 of Hb, direction stride dirH floats, and of the bf16 copy), zero forward hand-off counters
int gru_bf16_pack(var_ctx* c, hipStream_t s, const float* w_hh, const float* w_ih, long dirP, int nclips, int maxclips, float* Hb,
                  long dirH, void* ws);
int gru_bf16_step_fwd(var_ctx* c, hipStream_t s, const float* GI, float* Hb, const float* b_hh, long dirP, float* R, float* Z,
                      float* Nn, float* GHN, int nclips, int maxclips, int step, long dirGI, long dirH, long dirS, int save, void* ws);
int gru_bf16_step_bwd(var_ctx* c, hipStream_t s, float* DH, const float* Hb, const float* R, const float* Z, const float* Nn,
                      const float* GHN, float* DGI, float* DGH, int nclips, int maxclips, int step, int has_next, long dirGI,
                      long dirH, long dirS, long dirDGH, void* ws);
// the 73 steps of a pass in ONE persistent launch (workgroups of a clip slice hand the state over through L2 / HBM); 1 = the
// grid would not be resident on this device, take the per-step launches
int gru_bf16_seq_fwd(var_ctx* c, hipStream_t s, const float* GI, float* Hb, const float* b_hh, long dirP, float* R, float* Z, float* Nn,
                     float* GHN, int nclips, int maxclips, long dirGI, long dirH, long dirS, int save, void* ws, int drop_one);
int gru_bf16_seq_bwd(var_ctx* c, hipStream_t s, float* DH, const float* Hb, const float* R, const float* Z, const float* Nn,
                     const float* GHN, float* DGI, float* DGH, int nclips, int maxclips, long dirGI, long dirH, long dirS, long dirDGH,
                     void* ws, int store32);      // store32 = 0: DGI / DGH (fp32) are not written, only their bf16 copies
int gru_bf16_reset_timeout(var_ctx* c, hipStream_t s, int maxclips, void* ws);
// start of a step: a time-out left by the PREVIOUS step moves to the sticky words (count, last code) and the current word clears
int gru_bf16_step_begin(var_ctx* c, hipStream_t s, int maxclips, void* ws);
const unsigned* gru_bf16_timeout_ptr(int maxclips, void* ws);
int gru_bf16_poison_on_timeout(var_ctx* c, hipStream_t s, float* grads, int n, int maxclips, void* ws);
int gru_bf16_timeout_word(var_ctx* c, int maxclips, void* ws, unsigned* out);

// img_bf16.hip: 3x3 stride-1 convolutions of the iTHOR image branch (layers 2, 3) in the bf16 mode, forward and data gradient
long img_bf16_workspace_bytes();
int img_bf16_conv(var_ctx* c, hipStream_t s, int layer, int side, int dgrad, const float* x, const float* w, const float* bias,
                  const float* mask, float* y, float* csum, int* nparts, int B, void* ws, int prepacked);
// the fragment tables of layers 2-5 (forward and data gradient) in one launch; img_bf16_conv(..., prepacked = 1) then skips its own
int img_bf16_pack_all(var_ctx* c, hipStream_t s, const float* w2, const float* w3, const float* w4, const float* w5, void* ws);
int img_bf16_wgrad(var_ctx* c, hipStream_t s, int layer, int side, const float* x, const float* gy, float* dw, float* slab, int B);
int img_bf16_wgrad1(var_ctx* c, hipStream_t s, int side, const void* image, long bstride, const float* gy, float* dw, float* slab, int B);
