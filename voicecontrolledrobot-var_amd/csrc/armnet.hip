// RL actor-critic forward on gfx950 (BASELINE config 5 second half):
// models/RL/arm_RL_model.py:7-134 `armNet_VAR` (96x96 branch: 8 convolutions / 3 max pools -> 1152, the motor /
// image / sound MLPs, one GRU(128 -> 512) step through NNBase._forward_gru's acting path, models/ppo/model.py:116-121,
// fusion and the actor / critic trunks) followed by DiagGaussian's mean layer (models/ppo/distributions.py:65-84), i.e.
// everything of Policy.act up to the sampling.  Inference only (the PPO update stays in PyTorch).  Up to 64 images the
// convolutions run on the LDS-band kernels of c3f.h (filters re-packed per call inside conv 1's launch) and, up to 8 rows, the
// 22 Linear layers + GRU step on the one-launch chain below; larger batches take the gather-GEMM of gg.h layer by layer with
// the parameters in place in their state_dict() layouts.
#include <string.h>

#include <type_traits>

#include "gg.h"

namespace {
PH_DECL();
}
#include "c3f.h"
#ifdef VAR_PHASES
extern "C" int var_debug_phases_armchain(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

namespace {
constexpr int kCh[9] = {3, 32, 32, 64, 64, 128, 128, 256, 128};
constexpr int kRepr = 3, kRobot = 2, kRin = 128, kRh = 512, kAct = 128, kActions = 2, kFlat = 1152;

struct Lin { int w, b, in, out; };
struct ArmLayout {
    int g_wih, g_whh, g_bih, g_bhh;
    int cw[8], cb[8];
    Lin motor[3], cnn[2], im[2], im2, snd[3], fus[2], all[2], actor[2], critic[2], clin, mean;
    int logstd;
    int total;
};

ArmLayout make_layout() {
    ArmLayout L{};
    int o = 0;
    L.g_wih = o; o += 3 * kRh * kRin; L.g_whh = o; o += 3 * kRh * kRh; L.g_bih = o; o += 3 * kRh; L.g_bhh = o; o += 3 * kRh;
    for (int i = 0; i < 8; i++) { L.cw[i] = o; o += kCh[i + 1] * kCh[i] * 9; L.cb[i] = o; o += kCh[i + 1]; }
    auto lin = [&](int in, int out) { Lin l{o, o + in * out, in, out}; o += in * out + out; return l; };
    L.motor[0] = lin(kRepr + kRobot, 256); L.motor[1] = lin(256, 512); L.motor[2] = lin(512, 256);
    L.cnn[0] = lin(kFlat, 512); L.cnn[1] = lin(512, 256);
    L.im[0] = lin(256, 256); L.im[1] = lin(256, kRin);
    L.im2 = lin(kRh, 256);
    L.snd[0] = lin(kRepr, 128); L.snd[1] = lin(128, 256); L.snd[2] = lin(256, 256);
    L.fus[0] = lin(256, 512); L.fus[1] = lin(512, 256);
    L.all[0] = lin(256, 256); L.all[1] = lin(256, 128);
    L.actor[0] = lin(128, 128); L.actor[1] = lin(128, kAct);
    L.critic[0] = lin(128, 128); L.critic[1] = lin(128, 128);
    L.clin = lin(128, 1);
    L.mean = lin(kAct, kActions);
    L.logstd = o; o += kActions;
    L.total = o;
    return L;
}

struct arm_state {
    ArmLayout L;
    int maxB = 0;
    float* ws = nullptr;
    float *a[9] = {nullptr}, *p[4] = {nullptr};      // conv outputs 1..8, pooled maps 1..3
    float *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;   // (B,512) scratch rows
    float* slab = nullptr;
    float *flat_img = nullptr, *motor = nullptr, *sound = nullptr, *fusion = nullptr, *h0 = nullptr, *gi = nullptr, *gh = nullptr;
    float* chain = nullptr;        // the fused small-batch MLP chain's vectors (armnet_chain_kernel)
    unsigned* sync = nullptr;      // [1] finished workgroups, [2] epoch of the last launch that timed out, [3] epoch of the next launch,
                                   // [4] sticky: some launch timed out since the last var_armnet_clear_status
    bool drop_one = false;         // tests: the next chain launch runs one workgroup short (var_debug_armnet_drop_workgroup)
    c3f::f32x4* wpk = nullptr;     // conv 2..6 filters in MFMA A-fragment order (c3f.h), re-packed per forward
    c3f::PackDesc pack{};
};

// conv 2..6 of the 96x96 stack as band kernels (c3f.h): bands / channel groups chosen for ~192-256 workgroups at 8 images
using ArmC2 = c3f::Cfg<32, 32, 96, 4, 2, 1, true>;
using ArmC3 = c3f::Cfg<32, 64, 48, 6, 1, 1, false>;
using ArmC4 = c3f::Cfg<64, 64, 48, 6, 1, 1, true>;
using ArmC5 = c3f::Cfg<64, 128, 24, 8, 1, 1, false>;
using ArmC6 = c3f::Cfg<128, 128, 24, 4, 2, 2, true>;
using ArmC7 = c3f::SmallCfg<128, 256, 12, 2, 5>;
using ArmC8 = c3f::SmallCfg<256, 128, 5, 1, 3>;
constexpr int kBandMaxB = 64;      // beyond this the gather-GEMM's big tiles win

static __global__ void an_pool_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int H, int HP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = (int)(i % HP), py = (int)((i / HP) % HP);
    const long plane = i / ((long)HP * HP);
    const float* q = x + plane * H * H + (long)(2 * py) * H + 2 * px;
    y[i] = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[H], q[H + 1]));
}
// out = a + b (fusion sums), or out[b][:] = [u[b][:nu] | v[b][:nv]] (the motor input), or h * mask per row
static __global__ void an_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
static __global__ void an_cat_kernel(const float* __restrict__ u, int nu, const float* __restrict__ v, int nv, float* __restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (nu + nv)) return;
    const int b = i / (nu + nv), j = i - b * (nu + nv);
    out[i] = j < nu ? u[b * nu + j] : v[b * nv + j - nu];
}
static __global__ void an_mask_kernel(const float* __restrict__ h, const float* __restrict__ mask, float* __restrict__ out, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * H) out[i] = h[i] * mask[i / H];
}
// torch.nn.GRU cell (gate order r, z, n); gi / gh include their biases
static __global__ void an_gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                                          float* __restrict__ out, float* __restrict__ out2, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* a = gi + (long)b * 3 * H;
    const float* g = gh + (long)b * 3 * H;
    const float r = 1.f / (1.f + expf(-(a[j] + g[j])));
    const float z = 1.f / (1.f + expf(-(a[H + j] + g[H + j])));
    const float n = tanhf(a[2 * H + j] + r * g[2 * H + j]);
    const float v = (1.f - z) * n + z * h[i];
    out[i] = v;
    if (out2) out2[i] = v;
}


// ------------------------------------------------------------------------------------------------------------------
// Small batches (the RL stage's 8 envs): everything after the convolutions as ONE persistent launch.
// 22 Linear layers + the GRU step are 12 dependent stages of tiny products (8 rows x K <= 1152 x N <= 1536): as separate
// launches they cost ~4 us each whatever they compute (45 launches, 240 us of the 435-us forward).  Here kChainG workgroups
// stay resident; a stage's inputs (<= 57 KB for 8 rows) are staged into every workgroup's LDS, a wave takes blocks of four
// output features (lanes split K: coalesced 256-B reads of the weight rows in their state_dict() layout, 8 rows x 4
// outputs of partial sums per lane, a 31-shuffle butterfly leaves one (output, row) sum per lane).  Stages are NOT separated
// by a grid barrier (round 3's first form: sc1 stores drained, counter add, counter poll, sc1 loads -- four dependent trips
// through the fabric, ~8 us per stage): the handed-off vectors carry a tag, see st_pair() below.  The polls are bounded: a
// grid that is not resident sets sync[2], every workgroup stops waiting and the outputs are NaN.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kChainG = 128, kChainT = 256, kChainRows = 8, kChainNB = 4;
enum { IN_PLAIN = 0, IN_SUM, IN_CAT, IN_MASK, IN_GRU };
struct ChainJob { int w, b, K, N, kind, in0, in1, cat0, out, relu, wg0, nwg, out2; };   // out2: a second, plain copy of the output (or -1)
struct ChainStage { int job0, njobs; };
constexpr int kChainMaxJobs = 28, kChainMaxStages = 12, kChainBufs = 36;
struct ChainDesc {
    const float* P;
    float* buf[kChainBufs];
    ChainJob job[kChainMaxJobs];
    ChainStage stage[kChainMaxStages];
    int nstages, B, H;
    int b_hxs, b_mask, b_hout;           // buffer ids the GRU input kind needs besides in0 (gi) / in1 (gh)
    unsigned long long tagged;           // bit i: buffer i is handed over inside the launch as (value, tag) pairs
    unsigned* sync;                      // [1] finished workgroups, [2] epoch of the last timed-out launch, [3] epoch of the next launch, [4] sticky
};
typedef __attribute__((address_space(1))) float gf32;
typedef __attribute__((address_space(1))) unsigned gu32c;

// Hand-over without a barrier: every float a stage hands to a later one travels as an 8-byte (value, tag) pair written by ONE
// 64-bit sc1 store; the tag is the launch's epoch (a device counter the last workgroup of a launch bumps: captured graphs replay
// with frozen arguments).  A consumer polls its INPUT until every pair carries the epoch -- one trip through the fabric after the
// producer's store lands, instead of drain + counter add + counter poll + load (four), and a workgroup whose inputs are complete
// runs ahead of the others.  Every internal vector is written once per launch, so the epoch alone identifies it.
typedef __attribute__((address_space(1))) unsigned long long gu64c;
__device__ __forceinline__ void st_pair(float* buf, int e, float v, unsigned tag) {
    const unsigned long long q = (unsigned long long)__builtin_bit_cast(unsigned, v) | ((unsigned long long)tag << 32);
    __hip_atomic_store((gu64c*)(buf + 2 * e), q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr unsigned kChainSpinMax = 1u << 17;

__global__ void __launch_bounds__(kChainT) armnet_chain_kernel(ChainDesc D) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int dead_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = D.B, H = D.H;
    const unsigned epoch = __hip_atomic_load((gu32c*)(D.sync + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) dead_s = 0;
    bool alive = true;
    PHR_INIT(5, 0);
    // this workgroup's job of a stage (jobs own ranges of workgroups, sized by their weight volume)
    auto job_of = [&](int si) {
        const ChainStage S = D.stage[si];
        int q = 0;
#pragma unroll 1
        for (int t = 1; t < S.njobs; ++t) if ((int)blockIdx.x >= D.job[S.job0 + t].wg0) q = t;
        return D.job[S.job0 + q];
    };
    // The weights do not depend on the activations: the first eight k-chunks of this wave's first item of the NEXT stage (all
    // of them for every layer but the 1152-wide one) and its bias are requested before the stage's input is polled.
    // lane -> (output, row) after the butterfly: value o * 8 + r in the even lanes
    const int lidx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    using NB4 = std::integral_constant<int, kChainNB>;
    float pw[8][kChainNB], pbias = 0.f;
    auto load_w = [&](auto nbc, const ChainJob& J, int o0, int kc0, float (&wv)[8][kChainNB]) {
        constexpr int NB = decltype(nbc)::value;
        const float* W = D.P + J.w;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const int k = kc0 + c8 * 64 + lane;
#pragma unroll
            for (int o = 0; o < NB; ++o) {                                     // (unconditional, clamped: see the staging below)
                const int kk = k < J.K ? k : J.K - 1, oo = o0 + o < J.N ? o0 + o : J.N - 1;
                wv[c8][o] = W[(long)oo * J.K + kk];
            }
        }
    };
    auto prefetch = [&](const ChainJob& J) {
        const int jw = ((int)blockIdx.x - J.wg0) * (kChainT / 64) + wave;
        if (jw * kChainNB < J.N) {
            int oo;
            load_w(NB4{}, J, jw * kChainNB, 0, pw); oo = jw * kChainNB + lidx / kChainRows;
            pbias = D.P[J.b + (oo < J.N ? oo : J.N - 1)];
        }
    };
    auto give_up = [&]() {          // a producer never delivered (the grid is not resident): outputs NaN, the event is recorded
        if (*(volatile int*)&dead_s == 0) {
            __hip_atomic_store((gu32c*)(D.sync + 2), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // which launch (var_armnet_status)
            __hip_atomic_store((gu32c*)(D.sync + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // sticky
        }
        dead_s = 1;
    };
    // (once one wait of this workgroup has expired every later one gives up at its first turn: a grid that is not resident costs
    //  one bound, not one per poll)
    auto expired = [&](unsigned& spins) { return ++spins > kChainSpinMax || *(volatile int*)&dead_s != 0; };
    ChainJob J = job_of(0);
    bool mine = (int)blockIdx.x < J.wg0 + J.nwg;          // a stage may leave workgroups without a job (see split())
    if (mine) prefetch(J);
    __syncthreads();
#pragma unroll 1
    for (int si = 0; si < D.nstages; ++si) {
        if (!mine) {                                      // (workgroup-uniform; nothing to wait for: stages are not fenced)
            if (si + 1 == D.nstages) break;
            J = job_of(si + 1);
            mine = (int)blockIdx.x < J.wg0 + J.nwg;
            if (mine) prefetch(J);
            continue;
        }
        PHR(0);
        const int Kp = (J.K + 63) & ~63;
        float* xs = lds;          // two input buffers: the next stage is staged while slow waves still read this one
        const bool t0 = (D.tagged >> J.in0) & 1;          // handed-off input(s) -- IN_SUM pairs and the GRU's gi / gh are always both
        // ---- the job's input -> LDS [row][Kp], zero-padded ----
        {
            const float* a = D.buf[J.in0];
            const float* b2 = J.in1 >= 0 ? D.buf[J.in1] : nullptr;
            if (J.kind == IN_GRU) {
                // torch.nn.GRU cell (gate order r, z, n) from gi (in0) and gh (in1), both handed off: the new state is this layer's
                // input.  Two hidden units per thread and turn; a rolled loop on purpose (the code runs once per stage).
                const int kq = J.K >> 1, n2 = B * kq;
#pragma unroll 1
                for (int e = tid; e < n2; e += kChainT) {
                    const int r = e / kq, k = 2 * (e - r * kq);
                    const float2 hx = *(const float2*)(D.buf[D.b_hxs] + r * H + k);
                    const float hm = D.buf[D.b_mask][r];
                    unsigned long long q[6][2];
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int gte = 0; gte < 3; ++gte) {
                            const int el = r * 3 * H + gte * H + k;
                            q[gte][0] = __hip_atomic_load((gu64c*)(a + 2 * el), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            q[gte][1] = __hip_atomic_load((gu64c*)(a + 2 * el + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            q[3 + gte][0] = __hip_atomic_load((gu64c*)(b2 + 2 * el), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            q[3 + gte][1] = __hip_atomic_load((gu64c*)(b2 + 2 * el + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        unsigned bad = 0u;
#pragma unroll
                        for (int g = 0; g < 6; ++g) bad |= ((unsigned)(q[g][0] >> 32) ^ epoch) | ((unsigned)(q[g][1] >> 32) ^ epoch);
                        if (bad == 0u) break;
                        if (expired(spins)) { give_up(); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    float2 g3[6];
#pragma unroll
                    for (int g = 0; g < 6; ++g) g3[g] = float2{__builtin_bit_cast(float, (unsigned)q[g][0]), __builtin_bit_cast(float, (unsigned)q[g][1])};
                    const float gi_[2][3] = {{g3[0].x, g3[1].x, g3[2].x}, {g3[0].y, g3[1].y, g3[2].y}};
                    const float gh_[2][3] = {{g3[3].x, g3[4].x, g3[5].x}, {g3[3].y, g3[4].y, g3[5].y}};
                    const float hh[2] = {hx.x * hm, hx.y * hm};
                    float o[2];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const float rr = 1.f / (1.f + expf(-(gi_[c2][0] + gh_[c2][0])));
                        const float z = 1.f / (1.f + expf(-(gi_[c2][1] + gh_[c2][1])));
                        const float n = tanhf(gi_[c2][2] + rr * gh_[c2][2]);
                        o[c2] = (1.f - z) * n + z * hh[c2];
                    }
                    if (blockIdx.x == J.wg0) *(float2*)(D.buf[D.b_hout] + r * H + k) = float2{o[0], o[1]};       // rnn_hxs_out
                    *(float2*)(xs + r * Kp + k) = float2{o[0], o[1]};
                }
                for (int e = tid; e < (kChainRows - B) * Kp; e += kChainT) xs[B * Kp + e] = 0.f;
            } else if ((J.K & 3) == 0) {
                const int kq = J.K >> 1, n2 = kChainRows * kq;
                if (!t0) {
                    // kernel inputs (the convolutions' output, rnn_hxs * masks): nothing to poll.  All loads first, unconditional
                    // (addresses clamped into the buffers, values selected afterwards -- a load behind a per-lane condition becomes
                    // a branch with a wait of its own)
#pragma unroll 1
                    for (int e0 = tid; e0 < n2; e0 += 8 * kChainT) {
                        float2 v[8];
                        float hm[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            int e = e0 + i * kChainT;
                            e = e < n2 ? e : n2 - 1;
                            int r = e / kq;
                            const int k = 2 * (e - r * kq);
                            r = r < B ? r : B - 1;
                            v[i] = *(const float2*)(a + r * J.K + k);
                            hm[i] = J.kind == IN_MASK ? b2[r] : 1.f;
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int e = e0 + i * kChainT;
                            if (e >= n2) continue;
                            const int r = e / kq, k = 2 * (e - r * kq);
                            *(float2*)(xs + r * Kp + k) = r < B ? float2{v[i].x * hm[i], v[i].y * hm[i]} : float2{0.f, 0.f};
                        }
                    }
                } else {
                    // handed-off vectors ((value, tag) pairs; IN_SUM: two of them): a turn = 8 element pairs per thread, ALL of its
                    // 64-bit atomic loads issued back to back (hipcc keeps atomic loads in program order, so anything between two of
                    // them -- a tag compare, a branch -- makes every pair wait for the previous one), then the tags are checked; a
                    // turn that is not complete yet is simply taken again
                    const bool sum = J.kind == IN_SUM;
#pragma unroll 1
                    for (int e0 = tid; e0 < n2; e0 += 8 * kChainT) {
                        unsigned long long qa[8][2], qb[8][2];
                        int el[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            int e = e0 + i * kChainT;
                            e = e < n2 ? e : n2 - 1;
                            el[i] = 2 * e;                                    // (rows beyond the batch are handed over too)
                        }
                        unsigned spins = 0;
                        for (;;) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                qa[i][0] = __hip_atomic_load((gu64c*)(a + 2 * el[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                qa[i][1] = __hip_atomic_load((gu64c*)(a + 2 * el[i] + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            if (sum) {
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    qb[i][0] = __hip_atomic_load((gu64c*)(b2 + 2 * el[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    qb[i][1] = __hip_atomic_load((gu64c*)(b2 + 2 * el[i] + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; ++i) qb[i][0] = qb[i][1] = (unsigned long long)epoch << 32;      // value 0.f, tag ok
                            }
                            unsigned bad = 0u;
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                bad |= ((unsigned)(qa[i][0] >> 32) ^ epoch) | ((unsigned)(qa[i][1] >> 32) ^ epoch) |
                                       ((unsigned)(qb[i][0] >> 32) ^ epoch) | ((unsigned)(qb[i][1] >> 32) ^ epoch);
                            if (bad == 0u) break;
                            if (expired(spins)) { give_up(); break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int e = e0 + i * kChainT;
                            if (e >= n2) continue;
                            const int r = e / kq, k = 2 * (e - r * kq);
                            float2 o = float2{__builtin_bit_cast(float, (unsigned)qa[i][0]) + __builtin_bit_cast(float, (unsigned)qb[i][0]),
                                              __builtin_bit_cast(float, (unsigned)qa[i][1]) + __builtin_bit_cast(float, (unsigned)qb[i][1])};
                            if (r >= B) o = float2{0.f, 0.f};
                            *(float2*)(xs + r * Kp + k) = o;
                        }
                    }
                }
            } else {                                                            // the two tiny first layers (K = 5: [image_feat | robot_pose], K = 3): kernel inputs
                for (int e = tid; e < kChainRows * Kp; e += kChainT) {
                    const int r = e / Kp, k = e - r * Kp;
                    float v = 0.f;
                    if (r < B && k < J.K) {
                        if (J.kind == IN_CAT) v = k < J.cat0 ? a[r * J.cat0 + k] : b2[r * (J.K - J.cat0) + k - J.cat0];
                        else v = a[r * J.K + k];
                    }
                    xs[e] = v;
                }
            }
        }
        PHR(1);
        __syncthreads();
        alive = alive && dead_s == 0;
        PHR(2);
        // ---- items: blocks of kChainNB outputs, dealt to the waves of the job's workgroups.  (Blocks of ONE output where a job has a
        //      wave per output -- a quarter of the serial work per wave -- measured: 97.5 vs 94.5 us for the chain, the second
        //      instantiation's code does not pay for itself) ----
        auto items = [&](auto nbc) {
            constexpr int NB = decltype(nbc)::value;
            const int jw = ((int)blockIdx.x - J.wg0) * (kChainT / 64) + wave, jnw = J.nwg * (kChainT / 64);
            const int nitems = (J.N + NB - 1) / NB;
            const bool tout = (D.tagged >> J.out) & 1;
            bool first = true;
#pragma unroll 1
            for (int it = jw; it < nitems; it += jnw) {
                const int o0 = it * NB;
                float acc[NB][kChainRows];
#pragma unroll
                for (int o = 0; o < NB; ++o)
#pragma unroll
                    for (int r = 0; r < kChainRows; ++r) acc[o][r] = 0.f;
                float bias = pbias;
                if (!first) { const int oo = NB == 1 ? o0 : o0 + lidx / kChainRows; bias = D.P[J.b + (oo < J.N ? oo : J.N - 1)]; }
                // weight rows stream from HBM: eight k-chunks in flight per wave
#pragma unroll 1
                for (int kc0 = 0; kc0 < Kp; kc0 += 8 * 64) {
                    float wv[8][kChainNB];
                    if (first && kc0 == 0) {
#pragma unroll
                        for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
                            for (int o = 0; o < NB; ++o) wv[c8][o] = pw[c8][o];
                    } else load_w(nbc, J, o0, kc0, wv);
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) {
                        const int k = kc0 + c8 * 64 + lane;
                        if (kc0 + c8 * 64 < Kp) {
#pragma unroll
                            for (int r = 0; r < kChainRows; ++r) {
                                const float xv = xs[r * Kp + k];
#pragma unroll
                                for (int o = 0; o < NB; ++o) acc[o][r] = fmaf(wv[c8][o], xv, acc[o][r]);      // (padding k: xv == 0)
                            }
                        }
                    }
                }
                // butterfly over the 64 lanes: NB x 8 values -> one (output, row) sum per lane
                float v[NB * kChainRows];
#pragma unroll
                for (int o = 0; o < NB; ++o)
#pragma unroll
                    for (int r = 0; r < kChainRows; ++r) v[o * kChainRows + r] = acc[o][r];
                // (each step with compile-time constants: a loop over (half, bit) is not unrolled by hipcc, and the array then
                //  becomes 32-way select chains -- 6 K instructions.)  The two wide steps are gfx950's lane-swap instructions
                //  (v_permlane32_swap: lanes 32-63 of one register <-> lanes 0-31 of the other, v_permlane16_swap the same for the odd /
                //  even 16-lane rows): a step is one swap + one add per pair instead of two selects + a ds_bpermute + an add; xor 8 / 2 / 1
                //  are DPP moves (row_ror:8, quad_perm); only the two xor-4 exchanges go through the LDS crossbar.
                const auto fb = [](float x) { return __builtin_bit_cast(unsigned, x); };
                const auto bf = [](unsigned x) { return __builtin_bit_cast(float, x); };
#pragma unroll
                for (int i = 0; i < 16; ++i) { const auto q = __builtin_amdgcn_permlane32_swap(fb(v[i]), fb(v[i + 16]), false, false); v[i] = bf(q[0]) + bf(q[1]); }
#pragma unroll
                for (int i = 0; i < 8; ++i) { const auto q = __builtin_amdgcn_permlane16_swap(fb(v[i]), fb(v[i + 8]), false, false); v[i] = bf(q[0]) + bf(q[1]); }
#define CHAIN_FOLD(HALF, BIT, XCHG)                                                                   \
                {                                                                                     \
                    const bool up = (lane & (BIT)) != 0;                                              \
                    _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                             \
                        const float keep = up ? v[i + (HALF)] : v[i], give = up ? v[i] : v[i + (HALF)]; \
                        v[i] = keep + XCHG(give);                                                     \
                    }                                                                                 \
                }
#define X_ROR8(x) bf((unsigned)__builtin_amdgcn_update_dpp(0, (int)fb(x), 0x128, 0xf, 0xf, false))
#define X_XOR4(x) __shfl_xor(x, 4, 64)
#define X_XOR2(x) bf((unsigned)__builtin_amdgcn_update_dpp(0, (int)fb(x), 0x4e, 0xf, 0xf, false))
#define X_XOR1(x) bf((unsigned)__builtin_amdgcn_update_dpp(0, (int)fb(x), 0xb1, 0xf, 0xf, false))
                CHAIN_FOLD(4, 8, X_ROR8) CHAIN_FOLD(2, 4, X_XOR4) CHAIN_FOLD(1, 2, X_XOR2)
                float sum;
                int o, r;
                bool writer;
                sum = v[0] + X_XOR1(v[0]);
                o = lidx / kChainRows; r = lidx % kChainRows; writer = (lane & 1) == 0;
#undef X_ROR8
#undef X_XOR4
#undef X_XOR2
#undef X_XOR1
#undef CHAIN_FOLD
                first = false;
                if (writer && o0 + o < J.N) {
                    float y = sum + bias;
                    if (J.relu) y = fmaxf(y, 0.f);
                    if (!alive) y = __builtin_nanf("");
                    // rows beyond the batch are handed over too (zeros in, bias out): a consumer polls whole vectors
                    if (tout) st_pair(D.buf[J.out], r * J.N + o0 + o, y, epoch);
                    else if (r < B) D.buf[J.out][r * J.N + o0 + o] = y;
                    if (J.out2 >= 0 && r < B) D.buf[J.out2][r * J.N + o0 + o] = y;
                }
            }
        };
        items(NB4{});
        PHR(3);
        if (si + 1 == D.nstages) break;
        J = job_of(si + 1);
        mine = (int)blockIdx.x < J.wg0 + J.nwg;
        if (mine) prefetch(J);                                      // (in flight while the next input is polled)
        __syncthreads();
        PHR(4);
    }
    PHR_FLUSH();
    // the last workgroup to finish opens the next launch's epoch (its stores are drained first: nothing of this launch can
    // carry the new tag)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned d = __hip_atomic_fetch_add((gu32c*)(D.sync + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {
            __hip_atomic_store((gu32c*)(D.sync + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu32c*)(D.sync + 3), epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }
#define AN_CHECK(c) VAR_HIP_CHECK(c, hipGetLastError())
#define RUN(x) do { int r_ = (x); if (r_ != VAR_OK) return r_; } while (0)

constexpr long kSlab = 8L << 20;             // floats of split-K scratch

template <class G, bool U8>
int conv(var_ctx* c, hipStream_t s, arm_state* st, const ConvDims& d, const void* x, const float* w, const float* bias, float* y) {
    ConvFwdP<G, U8, false> p{};
    p.M = d.B * d.HO * d.WO; p.N = d.COUT; p.K = d.CIN * G::KHW;
    const long out = (long)p.M * p.N;
    p.nsplit = gg_small_split(((p.M + GG_MT - 1) / GG_MT) * ((p.N + 63) / 64), p.K, out, kSlab);
    p.d = d; p.x = x; p.w = w; p.bias = bias; p.y = y; p.slab = st->slab; p.sstride = out;
    RUN(gg_launch(c, s, p));
    if (p.nsplit > 1) {
        hipLaunchKernelGGL(gg_finish_kernel, g1(out), dim3(256), 0, s, y, st->slab, out, p.nsplit, out, bias, d.COUT, d.HO * d.WO, 1);
        AN_CHECK(c);
    }
    return VAR_OK;
}
int linear(var_ctx* c, hipStream_t s, arm_state* st, const float* P, const Lin& l, const float* X, float* Y, int rows, int relu) {
    const long out = (long)rows * l.out;
    const int ns = gg_small_split(((l.out + GG_MT - 1) / GG_MT) * ((rows + 63) / 64), l.in, out, kSlab);
    if (ns > 1) {
        DenseP<true, true, 2> p{};
        p.M = l.out; p.N = rows; p.K = l.in; p.nsplit = ns;
        p.A = P + l.w; p.sam = l.in; p.sak = 1; p.Bm = X; p.sbk = 1; p.sbn = l.in; p.C = st->slab; p.scm = 1; p.scn = l.out; p.sC = out;
        RUN(gg_launch(c, s, p));
        hipLaunchKernelGGL(gg_finish_kernel, g1(out), dim3(256), 0, s, Y, st->slab, out, ns, out, P + l.b, l.out, 1, relu);
        AN_CHECK(c);
        return VAR_OK;
    }
    DenseP<true, true, 0> p{};
    p.M = l.out; p.N = rows; p.K = l.in; p.nsplit = 1;
    p.A = P + l.w; p.sam = l.in; p.sak = 1; p.Bm = X; p.sbk = 1; p.sbn = l.in; p.C = Y; p.scm = 1; p.scn = l.out;
    p.bias = P + l.b; p.relu = relu;
    return gg_launch(c, s, p);
}
}  // namespace


namespace {
int chain_forward(var_ctx* c, hipStream_t s, arm_state* st, const float* P, const float* image_feat, const float* robot_pose,
                  const float* goal, const float* hxs, const float* masks, int B, float* value, float* actor_features, float* action_mean,
                  float* hxs_out) {
    const ArmLayout& L = st->L;
    ChainDesc D{};
    D.P = P; D.B = B; D.H = kRh; D.sync = st->sync;
    enum { A8, IMGF, POSE, GOAL, HXS, MASK, HOUT, VALUE, AFEAT, MEAN, CNN0, FLAT, M0, M1, MOTOR, S0, S1, SOUND, GH, IM0, X, F0, FUSION, GI, IMR,
           ALL0, ALL1, C0, C1, A0, AFT, NBUF };
    static_assert(NBUF <= kChainBufs, "buffer table");
    float* sc = st->chain;
    auto scratch = [&](int width) { float* p = sc; sc += kChainRows * width * 2; return p; };     // (value, tag) pairs
    D.buf[A8] = st->a[8]; D.buf[IMGF] = (float*)image_feat; D.buf[POSE] = (float*)robot_pose; D.buf[GOAL] = (float*)goal;
    D.buf[HXS] = (float*)hxs; D.buf[MASK] = (float*)masks; D.buf[HOUT] = hxs_out; D.buf[VALUE] = value; D.buf[AFEAT] = actor_features;
    D.buf[MEAN] = action_mean ? action_mean : scratch(kActions);
    const int widths[][2] = {{CNN0, 512}, {FLAT, 256}, {M0, 256}, {M1, 512}, {MOTOR, 256}, {S0, 128}, {S1, 256}, {SOUND, 256}, {GH, 3 * kRh},
                             {IM0, 256}, {X, kRin}, {F0, 512}, {FUSION, 256}, {GI, 3 * kRh}, {IMR, 256}, {ALL0, 256}, {ALL1, 128}, {C0, 128},
                             {C1, 128}, {A0, 128}, {AFT, kAct}};
    for (auto& wd : widths) { D.buf[wd[0]] = scratch(wd[1]); D.tagged |= 1ull << wd[0]; }      // handed over inside the launch
    D.b_hxs = HXS; D.b_mask = MASK; D.b_hout = HOUT;
    int nj = 0, ns = 0;
    long lds_max = 0, lds_cur = 0;
    auto stage = [&]() { D.stage[ns].job0 = nj; D.stage[ns].njobs = 0; lds_cur = 0; return ns++; };
    auto job = [&](const Lin& l, int kind, int in0, int in1, int cat0, int out, int relu, int out2 = -1) {
        D.job[nj] = ChainJob{l.w, l.b, l.in, l.out, kind, in0, in1, cat0, out, relu, 0, 0, out2};
        D.stage[ns - 1].njobs++;
        nj++;
        lds_cur = (long)kChainRows * ((l.in + 63) & ~63);              // a workgroup stages the input of ITS job only
        if (lds_cur > lds_max) lds_max = lds_cur;
    };
    // Workgroup ranges of a stage's jobs.  What a stage costs is its slowest wave's chain of dependent weight fetches (a block
    // of outputs = ceil(K / 512) batches of loads, ~2 us each from HBM; only a wave's first batch is requested ahead), NOT its
    // weight volume: sized by volume, the two tiny first layers (K = 5 and 3) got one workgroup each and their 64 / 32 blocks
    // took 35 us, a third of the whole chain, behind which everything else waited.  Greedy: every job starts with one
    // workgroup, the job with the longest per-wave chain gets the next one.
    auto split = [&]() {
        ChainStage& S = D.stage[ns - 1];
        auto chain_len = [&](const ChainJob& J, int nwg) {
            const int blocks = (J.N + kChainNB - 1) / kChainNB, waves = nwg * (kChainT / 64);
            return ((blocks + waves - 1) / waves) * ((J.K + 511) / 512);
        };
        int nwg[kChainMaxJobs] = {0};
        for (int q = 0; q < S.njobs; ++q) nwg[q] = 1;
        // (a job is never given more waves than it has blocks: an extra workgroup would only poll and stage the input once more
        //  -- the GRU stage's 196 KB per workgroup -- and the leftover workgroups skip the stage)
        auto full = [&](int q) { return nwg[q] * (kChainT / 64) >= (D.job[S.job0 + q].N + kChainNB - 1) / kChainNB; };
        for (int left = kChainG - S.njobs; left > 0; --left) {
            bool any = false;
            for (int q = 0; q < S.njobs; ++q) any = any || !full(q);
            if (!any) break;
            int worst = -1;
            for (int q = 0; q < S.njobs; ++q) {
                if (full(q)) continue;
                if (worst < 0) { worst = q; continue; }
                const int cq = chain_len(D.job[S.job0 + q], nwg[q]), cw = chain_len(D.job[S.job0 + worst], nwg[worst]);
                // ties: the job with more weight per workgroup (bandwidth is the second-order cost)
                if (cq > cw || (cq == cw && (double)D.job[S.job0 + q].K * D.job[S.job0 + q].N / nwg[q] >
                                                (double)D.job[S.job0 + worst].K * D.job[S.job0 + worst].N / nwg[worst])) worst = q;
            }
            ++nwg[worst];
        }
        int wg0 = 0;
        for (int q = 0; q < S.njobs; ++q) {
            ChainJob& J = D.job[S.job0 + q];
            J.wg0 = wg0; J.nwg = nwg[q];
            wg0 += nwg[q];
        }
    };
    const Lin ih{L.g_wih, L.g_bih, kRin, 3 * kRh}, hh{L.g_whh, L.g_bhh, kRh, 3 * kRh};
    stage(); job(L.cnn[0], IN_PLAIN, A8, -1, 0, CNN0, 1); job(L.motor[0], IN_CAT, IMGF, POSE, kRepr, M0, 1);
             job(L.snd[0], IN_PLAIN, GOAL, -1, 0, S0, 1); job(hh, IN_MASK, HXS, MASK, 0, GH, 0);
    split();
    stage(); job(L.cnn[1], IN_PLAIN, CNN0, -1, 0, FLAT, 1); job(L.motor[1], IN_PLAIN, M0, -1, 0, M1, 1); job(L.snd[1], IN_PLAIN, S0, -1, 0, S1, 1);
    split();
    stage(); job(L.motor[2], IN_PLAIN, M1, -1, 0, MOTOR, 1); job(L.snd[2], IN_PLAIN, S1, -1, 0, SOUND, 1);
    split();
    stage(); job(L.im[0], IN_SUM, FLAT, MOTOR, 0, IM0, 1); job(L.fus[0], IN_SUM, SOUND, FLAT, 0, F0, 1);
    split();
    stage(); job(L.im[1], IN_PLAIN, IM0, -1, 0, X, 1); job(L.fus[1], IN_PLAIN, F0, -1, 0, FUSION, 1);
    split();
    stage(); job(ih, IN_PLAIN, X, -1, 0, GI, 0);
    split();
    stage(); job(L.im2, IN_GRU, GI, GH, 0, IMR, 1);                                   // the GRU cell is its input transform
    split();
    stage(); job(L.all[0], IN_SUM, FUSION, IMR, 0, ALL0, 1);
    split();
    stage(); job(L.all[1], IN_PLAIN, ALL0, -1, 0, ALL1, 1);
    split();
    stage(); job(L.critic[0], IN_PLAIN, ALL1, -1, 0, C0, 1); job(L.actor[0], IN_PLAIN, ALL1, -1, 0, A0, 1);
    split();
    stage(); job(L.critic[1], IN_PLAIN, C0, -1, 0, C1, 1); job(L.actor[1], IN_PLAIN, A0, -1, 0, AFT, 1, AFEAT);
    split();
    stage(); job(L.clin, IN_PLAIN, C1, -1, 0, VALUE, 0); job(L.mean, IN_PLAIN, AFT, -1, 0, MEAN, 0);
    split();
    D.nstages = ns;
    if (ns > kChainMaxStages || nj > kChainMaxJobs || sc - st->chain > kChainRows * 16384) {
        VAR_SET_ERR(c, "armnet chain: table overflow");
        return VAR_ERR_ARG;
    }
    const int lds_bytes = (int)lds_max * 4;
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)armnet_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr |= var_dev_bit(c);
    }
    const int grid = kChainG - (st->drop_one ? 1 : 0);          // (one short: its outputs never arrive, every consumer's wait expires)
    st->drop_one = false;
    hipLaunchKernelGGL(armnet_chain_kernel, dim3(grid), dim3(kChainT), lds_bytes, s, D);
    AN_CHECK(c);
    return VAR_OK;
}
}  // namespace

void armnet_free(var_ctx* c) {
    arm_state* st = (arm_state*)c->arm;
    if (!st) return;
    if (st->ws) (void)hipFree(st->ws);
    delete st;
    c->arm = nullptr;
}

extern "C" {

int var_armnet_param_count(void) { return make_layout().total; }

int var_armnet_plan(var_ctx* c, int max_batch) {
    if (!c) return VAR_ERR_ARG;
    if (max_batch < 1 || max_batch > 4096) { VAR_SET_ERR(c, "var_armnet_plan: batch %d outside 1..4096", max_batch); return VAR_ERR_ARG; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    arm_state* st = (arm_state*)c->arm;
    if (st && st->maxB >= max_batch) return VAR_OK;
    if (st) {      // retire (do not free) the superseded workspace: a captured act() graph may still replay on it
        if (st->ws) { int rc = retire_block(c, st->ws); if (rc != VAR_OK) return rc; }
        delete st;
        c->arm = nullptr;
    }
    c->plan_gen++;
    st = new arm_state();
    c->arm = st;
    st->L = make_layout();
    st->maxB = max_batch;
    const long B = max_batch;
    const int side[9] = {96, 96, 96, 48, 48, 24, 24, 5, 3};          // output side of conv l
    long total = 0;
    auto take = [&](long n) { long o = total; total += (n + 63) & ~63L; return o; };
    long oa[9], op[4];
    for (int l = 1; l <= 8; ++l) oa[l] = take(B * kCh[l] * side[l] * side[l]);
    op[1] = take(B * 32 * 48 * 48); op[2] = take(B * 64 * 24 * 24); op[3] = take(B * 128 * 12 * 12);
    const long ot0 = take(B * 512), ot1 = take(B * 512), ot2 = take(B * 512), ot3 = take(B * 512);
    const long ofl = take(B * 256), omo = take(B * 256), osn = take(B * 256), ofu = take(B * 256), oh0 = take(B * kRh);
    const long ogi = take(B * 3 * kRh), ogh = take(B * 3 * kRh), oslab = take(kSlab);
    const long ochain = take(kChainRows * 16384), osync = take(64);
    {
        c3f::PackDesc& d = st->pack;
        d.n_layers = 7;
        int f4 = 0;
        for (int i = 0; i < 7; ++i) {
            const int l = i + 1;                          // conv l+1: kCh[l] -> kCh[l + 1]
            d.w_off[i] = st->L.cw[l]; d.cin[i] = kCh[l]; d.cout[i] = kCh[l + 1];
            d.wp_off[i] = f4; d.first[i] = f4;
            f4 += kCh[l] * kCh[l + 1] * 9 / 4;
        }
        d.first[7] = f4;
    }
    const long owpk = take(4L * st->pack.first[7]);
    VAR_HIP_CHECK(c, hipMalloc((void**)&st->ws, (size_t)total * sizeof(float)));
    float* w = st->ws;
    for (int l = 1; l <= 8; ++l) st->a[l] = w + oa[l];
    for (int l = 1; l <= 3; ++l) st->p[l] = w + op[l];
    st->t0 = w + ot0; st->t1 = w + ot1; st->t2 = w + ot2; st->t3 = w + ot3;
    st->flat_img = w + ofl; st->motor = w + omo; st->sound = w + osn; st->fusion = w + ofu; st->h0 = w + oh0;
    st->gi = w + ogi; st->gh = w + ogh; st->slab = w + oslab;
    st->chain = w + ochain; st->sync = (unsigned*)(w + osync);
    st->wpk = (c3f::f32x4*)(w + owpk);
    VAR_HIP_CHECK(c, hipMemset(st->chain, 0, (size_t)kChainRows * 16384 * sizeof(float)));       // no tag of any launch yet
    {
        const unsigned init[4] = {0u, 0u, 0u, 1u};                  // [3]: the first launch's epoch
        VAR_HIP_CHECK(c, hipMemset(st->sync, 0, 64 * sizeof(float)));
        VAR_HIP_CHECK(c, hipMemcpy(st->sync, init, sizeof(init), hipMemcpyHostToDevice));
    }
    return VAR_OK;
}

int var_armnet_forward(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8, long image_bstride,
                       const float* image_feat, const float* robot_pose, const float* goal_sound_feat,
                       const float* rnn_hxs, const float* masks, int B,
                       float* value, float* actor_features, float* action_mean, float* rnn_hxs_out) {
    if (!c) return VAR_ERR_ARG;
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    arm_state* st = (arm_state*)c->arm;
    if (!st || B > st->maxB) { VAR_SET_ERR(c, "var_armnet_forward: var_armnet_plan(%d) first", B); return VAR_ERR_PLAN; }
    if (!params || !image || !image_feat || !robot_pose || !goal_sound_feat || !rnn_hxs || !masks || !value ||
        !actor_features || !rnn_hxs_out || B < 1) {
        VAR_SET_ERR(c, "var_armnet_forward: NULL argument");
        return VAR_ERR_ARG;
    }
    {   // the small-batch chain reads rnn_hxs from every workgroup of its GRU stage while one of them writes rnn_hxs_out
        const char *a0 = (const char*)rnn_hxs, *b0 = (const char*)rnn_hxs_out;
        const size_t n = (size_t)B * kRh * sizeof(float);
        if (a0 < b0 + n && b0 < a0 + n) {
            VAR_SET_ERR(c, "var_armnet_forward: rnn_hxs_out overlaps rnn_hxs (an in-place state update is not supported)");
            return VAR_ERR_ARG;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const ArmLayout& L = st->L;
    const float* P = params;
    using S1 = Geo<3, 3, 1, 1, 1, 1>;
    using S2P0 = Geo<3, 3, 2, 2, 0, 0>;
    using S1P0 = Geo<3, 3, 1, 1, 0, 0>;
    auto dims = [&](int l, int hin, int stride, int pad) {
        return conv_dims(B, kCh[l - 1], hin, hin, kCh[l], 3, 3, stride, stride, pad, pad);
    };
    auto pool = [&](const float* x, float* y, int ch, int hin) -> int {
        const long n = (long)B * ch * (hin / 2) * (hin / 2);
        hipLaunchKernelGGL(an_pool_kernel, g1(n), dim3(256), 0, s, x, y, n, hin, hin / 2);
        AN_CHECK(c);
        return VAR_OK;
    };
    // imgCNN
    if (B <= kBandMaxB) {      // conv 1 and the filter pack of conv 2..8 in one launch (c3f.h)
        const c3f::PackDesc& d = st->pack;
        const int nconv = B * c3f::C1_BANDS, npack = (d.first[7] + 255) / 256;
        if (image_is_u8) hipLaunchKernelGGL(c3f::c1f_pack_kernel<true>, dim3(nconv + npack), dim3(256), 0, s, image, image_bstride, P, L.cw[0],
                                            L.cb[0], st->a[1], nconv, st->wpk, d);
        else hipLaunchKernelGGL(c3f::c1f_pack_kernel<false>, dim3(nconv + npack), dim3(256), 0, s, image, image_bstride, P, L.cw[0], L.cb[0],
                                st->a[1], nconv, st->wpk, d);
        AN_CHECK(c);
    } else {
        ConvDims d = dims(1, 96, 1, 1);
        d.xb = image_bstride;
        if (image_is_u8) RUN((conv<S1, true>(c, s, st, d, image, P + L.cw[0], P + L.cb[0], st->a[1])));
        else RUN((conv<S1, false>(c, s, st, d, image, P + L.cw[0], P + L.cb[0], st->a[1])));
    }
    if (B <= kBandMaxB) {
        const c3f::PackDesc& d = st->pack;
        RUN(c3f::launch<ArmC2>(c, s, st->a[1], st->wpk + d.wp_off[0], P + L.cb[1], st->p[1], B));
        RUN(c3f::launch<ArmC3>(c, s, st->p[1], st->wpk + d.wp_off[1], P + L.cb[2], st->a[3], B));
        RUN(c3f::launch<ArmC4>(c, s, st->a[3], st->wpk + d.wp_off[2], P + L.cb[3], st->p[2], B));
        RUN(c3f::launch<ArmC5>(c, s, st->p[2], st->wpk + d.wp_off[3], P + L.cb[4], st->a[5], B));
        RUN(c3f::launch<ArmC6>(c, s, st->a[5], st->wpk + d.wp_off[4], P + L.cb[5], st->p[3], B));
    } else {
        RUN((conv<S1, false>(c, s, st, dims(2, 96, 1, 1), st->a[1], P + L.cw[1], P + L.cb[1], st->a[2])));
        RUN(pool(st->a[2], st->p[1], 32, 96));
        RUN((conv<S1, false>(c, s, st, dims(3, 48, 1, 1), st->p[1], P + L.cw[2], P + L.cb[2], st->a[3])));
        RUN((conv<S1, false>(c, s, st, dims(4, 48, 1, 1), st->a[3], P + L.cw[3], P + L.cb[3], st->a[4])));
        RUN(pool(st->a[4], st->p[2], 64, 48));
        RUN((conv<S1, false>(c, s, st, dims(5, 24, 1, 1), st->p[2], P + L.cw[4], P + L.cb[4], st->a[5])));
        RUN((conv<S1, false>(c, s, st, dims(6, 24, 1, 1), st->a[5], P + L.cw[5], P + L.cb[5], st->a[6])));
        RUN(pool(st->a[6], st->p[3], 128, 24));
    }
    if (B <= kBandMaxB) {
        RUN(c3f::launch_small<ArmC7>(c, s, st->p[3], st->wpk + st->pack.wp_off[5], P + L.cb[6], st->a[7], B));
        RUN(c3f::launch_small<ArmC8>(c, s, st->a[7], st->wpk + st->pack.wp_off[6], P + L.cb[7], st->a[8], B));
    } else {
        RUN((conv<S2P0, false>(c, s, st, dims(7, 12, 2, 0), st->p[3], P + L.cw[6], P + L.cb[6], st->a[7])));
        RUN((conv<S1P0, false>(c, s, st, dims(8, 5, 1, 0), st->a[7], P + L.cw[7], P + L.cb[7], st->a[8])));
    }
    if (B <= kChainRows)      // the RL stage's batch: everything after the convolutions in one persistent launch
        return chain_forward(c, s, st, P, image_feat, robot_pose, goal_sound_feat, rnn_hxs, masks, B, value, actor_features, action_mean,
                             rnn_hxs_out);
    // image_flatten = cnnMlp(flatten)
    RUN(linear(c, s, st, P, L.cnn[0], st->a[8], st->t0, B, 1));
    RUN(linear(c, s, st, P, L.cnn[1], st->t0, st->flat_img, B, 1));
    // motor = motorMlp(cat(image_feat, robot_pose))
    hipLaunchKernelGGL(an_cat_kernel, g1(B * 5), dim3(256), 0, s, image_feat, kRepr, robot_pose, kRobot, st->t0, B);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.motor[0], st->t0, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.motor[1], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.motor[2], st->t2, st->motor, B, 1));
    // imageMotor = imgMotorMlp(image_flatten + motor)
    hipLaunchKernelGGL(an_add_kernel, g1(B * 256), dim3(256), 0, s, st->flat_img, st->motor, st->t0, B * 256);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.im[0], st->t0, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.im[1], st->t1, st->t2, B, 1));                       // (B,128)
    // one GRU step from hxs * masks (models/ppo/model.py:118-121)
    hipLaunchKernelGGL(an_mask_kernel, g1(B * kRh), dim3(256), 0, s, rnn_hxs, masks, st->h0, B, kRh);
    AN_CHECK(c);
    {
        const Lin ih{L.g_wih, L.g_bih, kRin, 3 * kRh}, hh{L.g_whh, L.g_bhh, kRh, 3 * kRh};
        RUN(linear(c, s, st, P, ih, st->t2, st->gi, B, 0));
        RUN(linear(c, s, st, P, hh, st->h0, st->gh, B, 0));
        hipLaunchKernelGGL(an_gru_cell_kernel, g1(B * kRh), dim3(256), 0, s, st->gi, st->gh, st->h0, st->t3, rnn_hxs_out, B, kRh);
        AN_CHECK(c);
    }
    RUN(linear(c, s, st, P, L.im2, st->t3, st->t0, B, 1));                         // imageMotorRnn (B,256)
    // sound, fusion
    RUN(linear(c, s, st, P, L.snd[0], goal_sound_feat, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.snd[1], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.snd[2], st->t2, st->sound, B, 1));
    hipLaunchKernelGGL(an_add_kernel, g1(B * 256), dim3(256), 0, s, st->sound, st->flat_img, st->t1, B * 256);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.fus[0], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.fus[1], st->t2, st->fusion, B, 1));
    hipLaunchKernelGGL(an_add_kernel, g1(B * 256), dim3(256), 0, s, st->fusion, st->t0, st->t1, B * 256);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.all[0], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.all[1], st->t2, st->t3, B, 1));                      // x (B,128)
    RUN(linear(c, s, st, P, L.critic[0], st->t3, st->t0, B, 1));
    RUN(linear(c, s, st, P, L.critic[1], st->t0, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.clin, st->t1, value, B, 0));
    RUN(linear(c, s, st, P, L.actor[0], st->t3, st->t0, B, 1));
    RUN(linear(c, s, st, P, L.actor[1], st->t0, actor_features, B, 1));
    if (action_mean) RUN(linear(c, s, st, P, L.mean, actor_features, action_mean, B, 0));
    return VAR_OK;
}

int var_armnet_status(var_ctx* c, unsigned* word) {
    if (!c) return VAR_ERR_ARG;
    arm_state* st = (arm_state*)c->arm;
    if (!st || !word) { VAR_SET_ERR(c, "var_armnet_status: var_armnet_plan first"); return VAR_ERR_PLAN; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    unsigned w[8] = {0};
    VAR_HIP_CHECK(c, hipMemcpy(w, st->sync, sizeof(w), hipMemcpyDeviceToHost));       // (blocking: behind the work already enqueued)
    // w[2]: epoch of the last launch whose waits expired; w[3]: epoch of the NEXT launch, so w[3] - 1 ran last
    *word = (w[2] != 0u && w[2] == w[3] - 1u) ? 1u : (w[4] ? 0x40000001u : 0u);
    return VAR_OK;
}

int var_armnet_clear_status(var_ctx* c) {
    if (!c) return VAR_ERR_ARG;
    arm_state* st = (arm_state*)c->arm;
    if (!st) { VAR_SET_ERR(c, "var_armnet_clear_status: var_armnet_plan first"); return VAR_ERR_PLAN; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    VAR_HIP_CHECK(c, hipDeviceSynchronize());
    const unsigned z = 0u;
    VAR_HIP_CHECK(c, hipMemcpy(st->sync + 2, &z, sizeof(z), hipMemcpyHostToDevice));
    VAR_HIP_CHECK(c, hipMemcpy(st->sync + 4, &z, sizeof(z), hipMemcpyHostToDevice));
    return VAR_OK;
}

int var_debug_armnet_drop_workgroup(var_ctx* c) {
    if (!c) return VAR_ERR_ARG;
    arm_state* st = (arm_state*)c->arm;
    if (!st) { VAR_SET_ERR(c, "var_debug_armnet_drop_workgroup: var_armnet_plan first"); return VAR_ERR_PLAN; }
    st->drop_one = true;
    return VAR_OK;
}

}  // extern "C"
