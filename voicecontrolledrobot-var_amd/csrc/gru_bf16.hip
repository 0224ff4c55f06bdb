// The recurrence of the iTHOR model's bidirectional GRU(448 -> 512) in its bf16 mode (models/pretext/ai2thor_pretext_model.py:
// 31-33; torch.nn.GRU gate order r, z, n): the recurrent product on v_mfma_f32_32x32x16_bf16 AND the gate arithmetic in one
// kernel, instead of a split-K gather-GEMM launch + its partial slabs + a gate kernel per step (32 us per step; the 146
// dependent steps were 25 % of the bf16 training step).  Two forms with identical arithmetic: one launch per time step
// (gru_step_*: 10.7 / 15.9 us per step with the kernel boundary), and the whole pass in one persistent launch
// (gru_seq_*: 6.9 / 11.9 us per step) whose workgroups hand the state over in memory -- see the note above gru_seq_fwd_kernel.
//
//   forward   gh[j][clip] = sum_k W_hh[g 512 + j][k] h[clip][k]         (3 gates x 32 hidden units x 64 clips per workgroup, K = 512)
//             r, z, n, h' from gi (the input projection), gh, b_hh; saved r, z, n, gh_n for the backward
//   backward  dh[clip][j] = DH[clip][j] + sum_g dgh_next[clip][g] W_hh[g][j]   (32 hidden units x 64 clips per workgroup, K = 1536)
//             then the gate derivatives of the step: DGI, DGH, DH = dh z
// Workgroup = (direction, 32 hidden units, 64 clips); wave = (32-clip block, half of K).  Both operands go from L2 straight
// to registers: W_hh re-packed once per pass into fragment order (bf16), the state / gate-gradient rows are fp32 in HBM
// (a lane's 8 k are 32 contiguous bytes) and rounded on the way in.  The two K halves meet in LDS; the gate arithmetic
// runs on the accumulator layout (lane = clip, 4 consecutive hidden units per register quad: 16-byte accesses).
#include "var_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned gu32;

PH_DECL();
#ifdef VAR_PHASES
extern "C" int var_debug_phases_gru(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#define PH_INIT3() unsigned long long ph_t = clock64(); const bool ph_on = blockIdx.x == 3 && blockIdx.y == 1 && blockIdx.z == 0 && threadIdx.x == 0
#else
#define PH_INIT3()
#endif

namespace {

constexpr int GH = 512, G3 = 1536, SEQ = 73;
constexpr int NJS = GH / 32;                           // hidden slices
constexpr long kWfBytes = 2L * NJS * 32 * 3 * 1024;    // forward fragments: [dir][js][ks 32][gate 3][lane] x 16 B
constexpr long kWbBytes = 2L * NJS * 96 * 1024;        // backward fragments: [dir][js][ks 96][lane] x 16 B

__device__ __forceinline__ unsigned bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return bf16_bits(a) | (bf16_bits(b) << 16); }
__device__ __forceinline__ bf16x8_t to_bf16x8(const float4 a, const float4 b) {
    u32x4_t v;
    v.x = pack2(a.x, a.y); v.y = pack2(a.z, a.w); v.z = pack2(b.x, b.y); v.w = pack2(b.z, b.w);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ u32x4_t wload(__amdgpu_buffer_rsrc_t r, int lane_off, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, byte_off, 0);
}
// gate non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each; the libm forms were a quarter of the
// step kernel); tanh(x) = 1 - 2 / (1 + e^{2x}) saturates correctly at both ends (e^{2x} -> 0 | inf)
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

// W_hh (dir stride dirP floats; [1536][512]) -> both fragment tables
// ... and, in the same launch (they were seven launches of a few us each): W_ih -> bf16, zero initial states (fp32 and bf16
// rows of step 0, both directions), zero forward hand-off counters
struct GruPrep { const float* w_ih; uint4* wih16; uint4* h16; uint4* hb; unsigned* cnt; long h16dir, hbdir; int nrow16, nrow32, ncnt; };
__global__ void __launch_bounds__(256) gru_pack_kernel(const float* __restrict__ w_hh, long dirP, uint4* __restrict__ wf,
                                                       uint4* __restrict__ wb, const GruPrep q) {
    int i = blockIdx.x * 256 + threadIdx.x;
    const int nf = 2 * NJS * 32 * 3 * 64, nb = 2 * NJS * 96 * 64;
    if (i >= nf + nb) {
        i -= nf + nb;
        const int n8 = G3 * 448 / 8;
        if (i < 2 * n8) {                                    // W_ih [dir][1536][448]
            const int d = i / n8, e = i - d * n8;
            const float4* x = (const float4*)(q.w_ih + d * dirP) + 2 * e;
            const float4 a = x[0], b = x[1];
            q.wih16[i] = make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
            return;
        }
        i -= 2 * n8;
        if (i < 2 * q.nrow16) { const int d = i / q.nrow16; q.h16[d * q.h16dir + (i - d * q.nrow16)] = make_uint4(0, 0, 0, 0); return; }
        i -= 2 * q.nrow16;
        if (i < 2 * q.nrow32) { const int d = i / q.nrow32; q.hb[d * q.hbdir + (i - d * q.nrow32)] = make_uint4(0, 0, 0, 0); return; }
        i -= 2 * q.nrow32;
        if (i < q.ncnt) q.cnt[i] = 0u;
        return;
    }
    unsigned v[8];
    if (i < nf) {
        const int lane = i & 63, gate = (i >> 6) % 3, ks = (i / 192) & 31, js = (i / (192 * 32)) % NJS, dir = i / (192 * 32 * NJS);
        const float* w = w_hh + dir * dirP + (long)(gate * GH + 32 * js + (lane & 31)) * GH + 16 * ks + 8 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf16_bits(w[e]);
        wf[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    } else if (i < nf + nb) {
        const int q = i - nf, lane = q & 63, ks = (q >> 6) % 96, js = (q / (64 * 96)) % NJS, dir = q / (64 * 96 * NJS);
        const float* w = w_hh + dir * dirP + (long)(16 * ks + 8 * (lane >> 5)) * GH + 32 * js + (lane & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf16_bits(w[(long)e * GH]);
        wb[q] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
}

// ---- forward step ----------------------------------------------------------------------------------------------
// 512 threads: wave = (32-clip block cbk, quarter of K kq), every operand load of the wave in flight at once.  The state
// is read as its bf16 copy H16 (written by the previous step next to the fp32 state).  The four K quarters meet in LDS;
// the gate phase re-maps the tile so that 8 consecutive lanes hold the 32 hidden units of ONE clip (128-byte rows of
// gi / r / z / n / h) instead of one clip per lane.
constexpr int RS = 66;                                  // LDS stride (16-byte slots) of a 64-lane block of accumulator quads
// The K partials meet in LDS as 16-byte quads (the 4 consecutive rows a lane holds per register quad): slot of
// (partial-and-gate block pg, register quad rq, clip column c, lane half h).  The rotation by 8 h and the block stride of 66
// make both sides conflict-free for ds_*_b128: a 16-lane group of the writing wave covers 16 consecutive columns of one
// half, one of the gate phase (two clips x 8 hidden quads) covers 2 rq + c + 8 h = sixteen different slots mod 16
// (ds_read_b32 moves a quarter of the bytes per clock: the 48 scalar reads per thread were 1 us of a 6.9 us step).
__device__ __forceinline__ int quad_slot(int pg, int rq, int c, int h) { return (pg * 4 + rq) * RS + ((c + 8 * h) & 31) + 32 * h; }

__global__ void __launch_bounds__(512) gru_step_fwd_kernel(const float* __restrict__ GI, float* __restrict__ Hb, uint2* __restrict__ H16,
                                                           const uint4* __restrict__ wf, const float* __restrict__ b_hh, long dirP,
                                                           float* __restrict__ R, float* __restrict__ Z, float* __restrict__ Nn,
                                                           float* __restrict__ GHN, int nclips, int step, long dirGI, long dirH,
                                                           long dirS, int save) {
    extern __shared__ __attribute__((aligned(16))) float red[];                       // quad_slot((cbk 4 + kq) 3 + gate, ..) x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, cbk = wave & 1, kq = wave >> 1;
    const int js = blockIdx.x, cs = blockIdx.y, dir = blockIdx.z;
    // the gate phase's operands first (thread = (clip_l = tid / 8, hidden quad jq = tid % 8)): their HBM latency runs
    // under the product
    const int jq = tid & 7, cl = tid >> 3, clip = 64 * cs + cl, clipc = min(clip, nclips - 1);
    const int t = dir ? SEQ - 1 - step : step, j = 32 * js + 4 * jq;
    const float* gi = GI + dir * dirGI + ((long)clipc * SEQ + t) * G3 + j;
    const float* bh = b_hh + dir * dirP + j;
    const float* hprev = Hb + dir * dirH + (long)step * nclips * GH + (long)clipc * GH + j;
    const float4 gr = *(const float4*)gi, gz = *(const float4*)(gi + GH), gn = *(const float4*)(gi + 2 * GH);
    const float4 br = *(const float4*)bh, bz = *(const float4*)(bh + GH), bn = *(const float4*)(bh + 2 * GH);
    const float4 hp = *(const float4*)hprev;
    __builtin_amdgcn_sched_barrier(0);
    {
        const int clip = min(64 * cs + 32 * cbk + (lane & 31), nclips - 1);
        const uint4* hsrc = (const uint4*)(H16 + (((long)dir * (SEQ + 1) + step) * nclips + clip) * (GH / 4)) + 16 * kq + h;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, (int)kWfBytes, 0x00020000);
        const int wbase = __builtin_amdgcn_readfirstlane(((dir * NJS + js) * 32 + 8 * kq) * 3 * 1024);   // (wave-uniform: a scalar offset)
        u32x4_t a[8][3];
        uint4 b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int g = 0; g < 3; ++g) a[i][g] = wload(wr, lane * 16, wbase + (i * 3 + g) * 1024);
            b[i] = hsrc[2 * i];
        }
        f32x16_t acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i][g]), __builtin_bit_cast(bf16x8_t, b[i]),
                                                                 acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                ((float4*)red)[quad_slot((cbk * 4 + kq) * 3 + g, rq, lane & 31, h)] =
                    make_float4(acc[g][4 * rq], acc[g][4 * rq + 1], acc[g][4 * rq + 2], acc[g][4 * rq + 3]);
    }
    __syncthreads();
    // gate phase: thread = (clip_l = tid / 8, hidden quad jq = tid % 8): units 32 js + 4 jq + 0..3, which the accumulators hold
    // in rows 4 (jq / 2) + e of lanes (clip_l % 32) + 32 (jq % 2)
    if (clip >= nclips) return;
    float gh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 p = ((const float4*)red)[quad_slot(((cl >> 5) * 4 + q) * 3 + g, jq >> 1, cl & 31, jq & 1)];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        gh[g][0] = v.x; gh[g][1] = v.y; gh[g][2] = v.z; gh[g][3] = v.w;
    }
    float* hnext = Hb + dir * dirH + (long)(step + 1) * nclips * GH + (long)clip * GH + j;
    const long so = dir * dirS + (long)step * nclips * GH + (long)clip * GH + j;
    float4 o, rr, zz, nn, gg;
#define GRU_LANE(c, e)                                                     \
    {                                                                      \
        const float r_ = sigmoidf_(gr.c + (gh[0][e] + br.c));              \
        const float z_ = sigmoidf_(gz.c + (gh[1][e] + bz.c));              \
        const float ghn_ = gh[2][e] + bn.c;                                \
        const float n_ = tanhf_(gn.c + r_ * ghn_);                          \
        o.c = (1.f - z_) * n_ + z_ * hp.c;                                 \
        rr.c = r_; zz.c = z_; nn.c = n_; gg.c = ghn_;                      \
    }
    GRU_LANE(x, 0) GRU_LANE(y, 1) GRU_LANE(z, 2) GRU_LANE(w, 3)
#undef GRU_LANE
    *(float4*)hnext = o;
    H16[(((long)dir * (SEQ + 1) + step + 1) * nclips + clip) * (GH / 4) + 8 * js + jq] = make_uint2(pack2(o.x, o.y), pack2(o.z, o.w));
    if (save) { *(float4*)(R + so) = rr; *(float4*)(Z + so) = zz; *(float4*)(Nn + so) = nn; *(float4*)(GHN + so) = gg; }
}

// ---- backward step ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) gru_step_bwd_kernel(float* __restrict__ DH, const float* __restrict__ Hb, const uint4* __restrict__ wb,
                                                           uint2* __restrict__ DG16, uint2* __restrict__ DGI16, const float* __restrict__ R,
                                                           const float* __restrict__ Z,
                                                           const float* __restrict__ Nn, const float* __restrict__ GHN,
                                                           float* __restrict__ DGI, float* __restrict__ DGH, int nclips, int step,
                                                           int has_next, long dirGI, long dirH, long dirS, long dirDGH) {
    extern __shared__ __attribute__((aligned(16))) float red[];                       // quad_slot(cbk 4 + kq, ..) x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, cbk = wave & 1, kq = wave >> 1;
    const int js = blockIdx.x, cs = blockIdx.y, dir = blockIdx.z;
    const int jq = tid & 7, cl = tid >> 3, clip = 64 * cs + cl, clipc = min(clip, nclips - 1);
    const int t = dir ? SEQ - 1 - step : step, j = 32 * js + 4 * jq;
    const long so = dir * dirS + (long)step * nclips * GH + (long)clipc * GH + j;
    const float* hprev = Hb + dir * dirH + (long)step * nclips * GH + (long)clipc * GH + j;
    float* dh = DH + (long)dir * nclips * GH + (long)clipc * GH + j;
    const float4 r4 = *(const float4*)(R + so), z4 = *(const float4*)(Z + so), n4 = *(const float4*)(Nn + so);
    const float4 g4 = *(const float4*)(GHN + so), hp = *(const float4*)hprev, d4 = *(const float4*)dh;
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {                                      // uniform
        const int clip = min(64 * cs + 32 * cbk + (lane & 31), nclips - 1);
        const uint4* gsrc = (const uint4*)(DG16 + (((long)dir * SEQ + step + 1) * nclips + clip) * (G3 / 4)) + 48 * kq + h;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)kWbBytes, 0x00020000);
        const int wbase = __builtin_amdgcn_readfirstlane(((dir * NJS + js) * 96 + 24 * kq) * 1024);
        u32x4_t a[24];
        uint4 b[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) { a[i] = wload(wr, lane * 16, wbase + i * 1024); b[i] = gsrc[2 * i]; }
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 24; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[i]), acc, 0, 0, 0);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            ((float4*)red)[quad_slot(cbk * 4 + kq, rq, lane & 31, h)] = make_float4(acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
        __syncthreads();
    }
    if (clip >= nclips) return;
    float dp[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_next) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 p = ((const float4*)red)[quad_slot((cl >> 5) * 4 + q, jq >> 1, cl & 31, jq & 1)];
            dp[0] += p.x; dp[1] += p.y; dp[2] += p.z; dp[3] += p.w;
        }
    }
    float* dgi = DGI + dir * dirGI + ((long)clip * SEQ + t) * G3 + j;
    float* dgh = DGH + dir * dirDGH + ((long)step * nclips + clip) * G3 + j;
    float4 dr, dz, dn, dnr, dd;
#define GRU_LANE(c, e)                                                     \
    {                                                                      \
        const float dh_ = d4.c + dp[e];                                    \
        const float dn_ = dh_ * (1.f - z4.c) * (1.f - n4.c * n4.c);        \
        dz.c = dh_ * (hp.c - n4.c) * z4.c * (1.f - z4.c);                  \
        dr.c = dn_ * g4.c * r4.c * (1.f - r4.c);                           \
        dn.c = dn_; dnr.c = dn_ * r4.c; dd.c = dh_ * z4.c;                 \
    }
    GRU_LANE(x, 0) GRU_LANE(y, 1) GRU_LANE(z, 2) GRU_LANE(w, 3)
#undef GRU_LANE
    *(float4*)dgi = dr; *(float4*)(dgi + GH) = dz; *(float4*)(dgi + 2 * GH) = dn;
    *(float4*)dgh = dr; *(float4*)(dgh + GH) = dz; *(float4*)(dgh + 2 * GH) = dnr;
    *(float4*)dh = dd;
    // bf16 copies of both gate-gradient rows: the next step's product reads DG16, the dW / dX products read both
    uint2* g16 = DG16 + (((long)dir * SEQ + step) * nclips + clip) * (G3 / 4) + 8 * js + jq;
    uint2* i16 = DGI16 + ((long)dir * SEQ * nclips + (long)clip * SEQ + t) * (G3 / 4) + 8 * js + jq;
    const uint2 pr = make_uint2(pack2(dr.x, dr.y), pack2(dr.z, dr.w)), pz = make_uint2(pack2(dz.x, dz.y), pack2(dz.z, dz.w));
    g16[0] = pr; g16[GH / 4] = pz; g16[2 * GH / 4] = make_uint2(pack2(dnr.x, dnr.y), pack2(dnr.z, dnr.w));
    i16[0] = pr; i16[GH / 4] = pz; i16[2 * GH / 4] = make_uint2(pack2(dn.x, dn.y), pack2(dn.z, dn.w));
}

// ---- the whole sequence in one launch -----------------------------------------------------------------------------
// A step couples the 512 hidden units of ONE clip, never two clips: the 16 workgroups (hidden slices) of a (direction,
// 64-clip slice) form a group that only has to wait for itself.  One persistent launch per pass keeps W_hh's fragments
// in registers for all 73 steps and replaces the kernel boundary by a 16-arrival counter per (group, step).  The
// hand-off follows the guide's write-through form (MI355X_MICROARCH.md, inter-workgroup visibility, first table row):
// every handed-off byte (the bf16 state / gate-gradient rows) is stored sc1 and drained by its wave (s_waitcnt vmcnt(0))
// before the workgroup's barrier, ONE lane then adds to the counter (agent scope), ONE lane of each consumer polls it
// with sc1 loads, the workgroup's barrier follows, and every load of those bytes is an sc1 buffer load to registers.
// Everything else a step stores (fp32 states, saved gates, DGI / DGH) is read by later launches only.  The spin is
// bounded: a group that never completes (a grid that is not resident) sets the time-out word and every workgroup leaves.
constexpr unsigned kSpinMax = 1u << 18;
constexpr int kSc1 = 16;                                 // buffer aux bit: sc1

__device__ __forceinline__ int wait_count(unsigned* cnt, unsigned want, unsigned* tmo, unsigned code) {
    for (unsigned spins = 0; spins < kSpinMax; ++spins) {
        if (__hip_atomic_load((gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return 1;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store((gu32*)tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
}

__global__ void __launch_bounds__(512) gru_seq_fwd_kernel(const float* __restrict__ GI, float* __restrict__ Hb, uint2* H16,
                                                          const uint4* __restrict__ wf, const float* __restrict__ b_hh, long dirP,
                                                          float* __restrict__ R, float* __restrict__ Z, float* __restrict__ Nn,
                                                          float* __restrict__ GHN, int nclips, long dirGI, long dirH, long dirS,
                                                          int save, unsigned* cnt0, unsigned* tmo) {
    extern __shared__ __attribute__((aligned(16))) float red[];                       // quad_slot((cbk 4 + kq) 3 + gate, ..) x 16 B | go
    int* go = (int*)(red + 2 * 4 * 3 * 16 * RS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, cbk = wave & 1, kq = wave >> 1;
    const int js = blockIdx.x, cs = blockIdx.y, dir = blockIdx.z;
    const int jq = tid & 7, cl = tid >> 3, clip = 64 * cs + cl, clipc = min(clip, nclips - 1);
    const int j = 32 * js + 4 * jq;
    unsigned* cnt = cnt0 + (dir * gridDim.y + cs) * (SEQ + 1);
    const float* bh = b_hh + dir * dirP + j;
    const float4 br = *(const float4*)bh, bz = *(const float4*)(bh + GH), bn = *(const float4*)(bh + 2 * GH);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, (int)kWfBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void*)H16, 0, 2 * (SEQ + 1) * nclips * GH * 2, 0x00020000);
    const int wbase = __builtin_amdgcn_readfirstlane(((dir * NJS + js) * 32 + 8 * kq) * 3 * 1024);   // (wave-uniform: a scalar offset)
    u32x4_t a[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 3; ++g) a[i][g] = wload(wr, lane * 16, wbase + (i * 3 + g) * 1024);
    const int ldoff = min(64 * cs + 32 * cbk + (lane & 31), nclips - 1) * (GH * 2) + (16 * kq + h) * 16;   // this lane's operand bytes in a step's rows
    const int stoff = clipc * (GH * 2) + (8 * js + jq) * 8;                                                  // its bf16 output
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
    PH_INIT3();
    for (int step = 0; step < SEQ; ++step) {
        PH(0);
        const int t = dir ? SEQ - 1 - step : step;
        const float* gi = GI + dir * dirGI + ((long)clipc * SEQ + t) * G3 + j;
        const float4 gr = *(const float4*)gi, gz = *(const float4*)(gi + GH), gn = *(const float4*)(gi + 2 * GH);
        if (step) {
            if (tid == 0) *go = wait_count(cnt + step, NJS, tmo, 1 + step);
            __syncthreads();
            if (!*go) break;
            PH(1);
        }
        const int rows = ((dir * (SEQ + 1) + step) * nclips) * (GH * 2);
        u32x4_t b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = __builtin_amdgcn_raw_buffer_load_b128(hr, ldoff + 32 * i, rows, kSc1);
#ifdef VAR_PHASES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PH(2);
#endif
        f32x16_t acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i][g]), __builtin_bit_cast(bf16x8_t, b[i]),
                                                                 acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                ((float4*)red)[quad_slot((cbk * 4 + kq) * 3 + g, rq, lane & 31, h)] =
                    make_float4(acc[g][4 * rq], acc[g][4 * rq + 1], acc[g][4 * rq + 2], acc[g][4 * rq + 3]);
        PH(3);
        __syncthreads();
        PH(4);
        float4 o, rr, zz, nn, gg;
        if (clip < nclips) {
            float gh[3][4];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 p = ((const float4*)red)[quad_slot(((cl >> 5) * 4 + q) * 3 + g, jq >> 1, cl & 31, jq & 1)];
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
                gh[g][0] = v.x; gh[g][1] = v.y; gh[g][2] = v.z; gh[g][3] = v.w;
            }

#define GRU_LANE(c, e)                                                     \
    {                                                                      \
        const float r_ = sigmoidf_(gr.c + (gh[0][e] + br.c));              \
        const float z_ = sigmoidf_(gz.c + (gh[1][e] + bz.c));              \
        const float ghn_ = gh[2][e] + bn.c;                                \
        const float n_ = tanhf_(gn.c + r_ * ghn_);                          \
        o.c = (1.f - z_) * n_ + z_ * hp.c;                                 \
        rr.c = r_; zz.c = z_; nn.c = n_; gg.c = ghn_;                      \
    }
            GRU_LANE(x, 0) GRU_LANE(y, 1) GRU_LANE(z, 2) GRU_LANE(w, 3)
#undef GRU_LANE
            u32x2_t o16;
            o16.x = pack2(o.x, o.y); o16.y = pack2(o.z, o.w);
            __builtin_amdgcn_raw_buffer_store_b64(o16, hr, stoff, rows + nclips * (GH * 2), kSc1);
            hp = o;
        }
        PH(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PH(6);
        __syncthreads();
        PH(7);
        if (tid == 0) __hip_atomic_fetch_add((gu32*)(cnt + step + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // what only LATER launches read (the fp32 state, the saved gates) is stored behind the hand-off: the drain above then
        // waits for the one bf16 row the next step needs, not for five more 16-byte stores per lane
        if (clip < nclips) {
            float* hnext = Hb + dir * dirH + (long)(step + 1) * nclips * GH + (long)clip * GH + j;
            const long so = dir * dirS + (long)step * nclips * GH + (long)clip * GH + j;
            *(float4*)hnext = o;
            if (save) { *(float4*)(R + so) = rr; *(float4*)(Z + so) = zz; *(float4*)(Nn + so) = nn; *(float4*)(GHN + so) = gg; }
        }
    }
}

__global__ void __launch_bounds__(512) gru_seq_bwd_kernel(float* __restrict__ DH, const float* __restrict__ Hb, const uint4* __restrict__ wb,
                                                          uint2* DG16, uint2* __restrict__ DGI16, const float* __restrict__ R,
                                                          const float* __restrict__ Z, const float* __restrict__ Nn,
                                                          const float* __restrict__ GHN, float* __restrict__ DGI, float* __restrict__ DGH,
                                                          int nclips, long dirGI, long dirH, long dirS, long dirDGH, unsigned* cnt0,
                                                          unsigned* tmo, float* __restrict__ bias_part, int store32) {
    extern __shared__ __attribute__((aligned(16))) float red[];                       // quad_slot(cbk 4 + kq, ..) x 16 B | go
    int* go = (int*)(red + 2 * 4 * 16 * RS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, cbk = wave & 1, kq = wave >> 1;
    const int js = blockIdx.x, cs = blockIdx.y, dir = blockIdx.z;
    const int jq = tid & 7, cl = tid >> 3, clip = 64 * cs + cl, clipc = min(clip, nclips - 1);
    const int j = 32 * js + 4 * jq;
    unsigned* cnt = cnt0 + (dir * gridDim.y + cs) * (SEQ + 1);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)kWbBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)DG16, 0, 2 * SEQ * nclips * G3 * 2, 0x00020000);
    const int wbase = __builtin_amdgcn_readfirstlane(((dir * NJS + js) * 96 + 24 * kq) * 1024);
    u32x4_t a[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) a[i] = wload(wr, lane * 16, wbase + i * 1024);
    const int ldoff = min(64 * cs + 32 * cbk + (lane & 31), nclips - 1) * (G3 * 2) + (48 * kq + h) * 16;
    const int stoff = clipc * (G3 * 2) + (8 * js + jq) * 8;
    float* dh = DH + (long)dir * nclips * GH + (long)clipc * GH + j;
    float4 d4 = *(const float4*)dh;
    float4 sr = make_float4(0.f, 0.f, 0.f, 0.f), sz = sr, sn = sr, snr = sr;     // this thread's share of the bias gradients
    PH_INIT3();
    for (int step = SEQ - 1; step >= 0; --step) {
        PH(10);
        const int t = dir ? SEQ - 1 - step : step;
        const long so = dir * dirS + (long)step * nclips * GH + (long)clipc * GH + j;
        const float* hprev = Hb + dir * dirH + (long)step * nclips * GH + (long)clipc * GH + j;
        const float4 r4 = *(const float4*)(R + so), z4 = *(const float4*)(Z + so), n4 = *(const float4*)(Nn + so);
        const float4 g4 = *(const float4*)(GHN + so), hp = *(const float4*)hprev;
        const int rows = ((dir * SEQ + step) * nclips) * (G3 * 2);
        float dp[4] = {0.f, 0.f, 0.f, 0.f};
        if (step < SEQ - 1) {
            if (tid == 0) *go = wait_count(cnt + step + 1, NJS, tmo, 101 + step);
            __syncthreads();
            if (!*go) break;
            PH(11);
            u32x4_t b[24];
#pragma unroll
            for (int i = 0; i < 24; ++i) b[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, ldoff + 32 * i, rows + nclips * (G3 * 2), kSc1);
#ifdef VAR_PHASES
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PH(12);
#endif
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 24; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[i]), acc, 0, 0, 0);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                ((float4*)red)[quad_slot(cbk * 4 + kq, rq, lane & 31, h)] = make_float4(acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
            PH(13);
            __syncthreads();
            PH(14);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 p = ((const float4*)red)[quad_slot((cl >> 5) * 4 + q, jq >> 1, cl & 31, jq & 1)];
                dp[0] += p.x; dp[1] += p.y; dp[2] += p.z; dp[3] += p.w;
            }
        }
        float4 dr, dz, dn, dnr;
        u32x2_t pr, pz, pn;
        if (clip < nclips) {
            float4 dd;
#define GRU_LANE(c, e)                                                     \
    {                                                                      \
        const float dh_ = d4.c + dp[e];                                    \
        const float dn_ = dh_ * (1.f - z4.c) * (1.f - n4.c * n4.c);        \
        dz.c = dh_ * (hp.c - n4.c) * z4.c * (1.f - z4.c);                  \
        dr.c = dn_ * g4.c * r4.c * (1.f - r4.c);                           \
        dn.c = dn_; dnr.c = dn_ * r4.c; dd.c = dh_ * z4.c;                 \
    }
            GRU_LANE(x, 0) GRU_LANE(y, 1) GRU_LANE(z, 2) GRU_LANE(w, 3)
#undef GRU_LANE
            pr.x = pack2(dr.x, dr.y); pr.y = pack2(dr.z, dr.w);
            pz.x = pack2(dz.x, dz.y); pz.y = pack2(dz.z, dz.w);
            pn.x = pack2(dnr.x, dnr.y); pn.y = pack2(dnr.z, dnr.w);
            __builtin_amdgcn_raw_buffer_store_b64(pr, gr, stoff, rows, kSc1);
            __builtin_amdgcn_raw_buffer_store_b64(pz, gr, stoff + GH * 2, rows, kSc1);
            __builtin_amdgcn_raw_buffer_store_b64(pn, gr, stoff + 2 * GH * 2, rows, kSc1);
            d4 = dd;
            sr.x += dr.x; sr.y += dr.y; sr.z += dr.z; sr.w += dr.w;
            sz.x += dz.x; sz.y += dz.y; sz.z += dz.z; sz.w += dz.w;
            sn.x += dn.x; sn.y += dn.y; sn.z += dn.z; sn.w += dn.w;
            snr.x += dnr.x; snr.y += dnr.y; snr.z += dnr.z; snr.w += dnr.w;
        }
        PH(15);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PH(16);
        __syncthreads();
        PH(17);
        if (tid == 0) __hip_atomic_fetch_add((gu32*)(cnt + step), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (as in the forward pass: what later launches read is stored behind the hand-off)
        if (clip < nclips) {
            if (store32) {      // (the fp32 arrays: nothing reads them when the products take the bf16 copies and the bias sums are formed here)
                float* dgi = DGI + dir * dirGI + ((long)clip * SEQ + t) * G3 + j;
                float* dgh = DGH + dir * dirDGH + ((long)step * nclips + clip) * G3 + j;
                *(float4*)dgi = dr; *(float4*)(dgi + GH) = dz; *(float4*)(dgi + 2 * GH) = dn;
                *(float4*)dgh = dr; *(float4*)(dgh + GH) = dz; *(float4*)(dgh + 2 * GH) = dnr;
            }
            uint2* i16 = DGI16 + ((long)dir * SEQ * nclips + (long)clip * SEQ + t) * (G3 / 4) + 8 * js + jq;
            i16[0] = make_uint2(pr.x, pr.y); i16[GH / 4] = make_uint2(pz.x, pz.y);
            i16[2 * GH / 4] = make_uint2(pack2(dn.x, dn.y), pack2(dn.z, dn.w));
        }
    }
    if (clip < nclips) *(float4*)dh = d4;
    // bias gradients: b_ih = sum over (clip, t) of (dr, dz, dn), b_hh of (dr, dz, dn r) -- this workgroup's 64 clips summed
    // in a fixed order (8 clips of a wave by lane exchange, the 8 waves through LDS); the slices of a direction are folded
    // by the caller.  (A launch that timed out leaves partial sums: the caller's time-out check poisons the gradient.)
    float4 v[4] = {sr, sz, sn, snr};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            v[k].x += __shfl_xor(v[k].x, m); v[k].y += __shfl_xor(v[k].y, m);
            v[k].z += __shfl_xor(v[k].z, m); v[k].w += __shfl_xor(v[k].w, m);
        }
    float4* bs = (float4*)red;                           // [wave 8][kind 4][jq 8]
    __syncthreads();
    if (lane < 8)
#pragma unroll
        for (int k = 0; k < 4; ++k) bs[(wave * 4 + k) * 8 + lane] = v[k];
    __syncthreads();
    if (tid < 32) {
        const int k = tid >> 3, q = tid & 7;
        float4 t = bs[k * 8 + q];
        for (int w = 1; w < 8; ++w) { const float4 u = bs[(w * 4 + k) * 8 + q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        // [dir][slice][b_ih: r z n | b_hh: r z n][512]
        float* o = bias_part + ((long)(dir * gridDim.y + cs) * 6) * GH + 32 * js + 4 * q;
        if (k < 3) *(float4*)(o + k * GH) = t;
        if (k < 2) *(float4*)(o + (3 + k) * GH) = t;
        if (k == 3) *(float4*)(o + 5 * GH) = t;
    }
}

// a timed-out sequence kernel has left partial results: make that loud in the gradient (NaN) rather than silent
__global__ void gru_poison_kernel(const unsigned* __restrict__ tmo, float* __restrict__ g, int n) {
    if (*tmo == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = __uint_as_float(0x7fc00000u);
}

// fp32 -> bf16, 8 elements per thread (n % 8 == 0)
__global__ void __launch_bounds__(256) to_bf16_kernel(const float4* __restrict__ x, uint4* __restrict__ y, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = x[2 * i], b = x[2 * i + 1];
    y[i] = make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
}

}  // namespace

// workspace: [Wf | Wb | H16 (dir, step 0..73, clip, 512) | DGH16 (dir, step, clip, 1536) | DGI16 (dir, clip*73+t, 1536) | X16 (clip*73+t, 448)], bf16
static inline long h16_bytes(int mc) { return 2L * (SEQ + 1) * mc * GH * 2; }
static inline long dg16_bytes(int mc) { return 2L * SEQ * mc * G3 * 2; }
static inline long x16_bytes(int mc) { return (long)SEQ * mc * 448 * 2; }
constexpr long kWih16Bytes = 2L * G3 * 448 * 2;          // W_ih as bf16, [dir][1536][448]
static inline long bias_part_bytes(int mc) { return 2L * ((mc + 63) / 64) * 6 * GH * 4; }   // [dir][clip slice][6][512] fp32
// hand-off words of the sequence kernels: [time-out word, 3 pad | forward counters (dir, clip slice, step 0..73) | backward counters]
static inline int sync_counters(int mc) { return 2 * ((mc + 63) / 64) * (SEQ + 1); }
static inline long sync_bytes(int mc) { return ((4 + 2L * sync_counters(mc)) * 4 + 255) & ~255L; }
long gru_bf16_workspace_bytes(int mc) {
    return kWfBytes + kWbBytes + h16_bytes(mc) + 2 * dg16_bytes(mc) + ((x16_bytes(mc) + 255) & ~255L) + sync_bytes(mc) + kWih16Bytes + bias_part_bytes(mc) + 256;
}
void* gru_bf16_h16(void* ws) { return (char*)ws + kWfBytes + kWbBytes; }
void* gru_bf16_dgh16(void* ws, int mc) { return (char*)gru_bf16_h16(ws) + h16_bytes(mc); }
void* gru_bf16_dgi16(void* ws, int mc) { return (char*)gru_bf16_dgh16(ws, mc) + dg16_bytes(mc); }
void* gru_bf16_x16(void* ws, int mc) { return (char*)gru_bf16_dgi16(ws, mc) + dg16_bytes(mc); }
static unsigned* gru_sync(void* ws, int mc) { return (unsigned*)((char*)gru_bf16_x16(ws, mc) + ((x16_bytes(mc) + 255) & ~255L)); }
void* gru_bf16_wih16(void* ws, int mc) { return (char*)gru_sync(ws, mc) + sync_bytes(mc); }
float* gru_bf16_bias_part(void* ws, int mc) { return (float*)((char*)gru_bf16_wih16(ws, mc) + kWih16Bytes); }
constexpr int kFwdLds = 2 * 4 * 3 * 16 * RS * 4, kBwdLds = 2 * 4 * 16 * RS * 4;

// the GRU's input sequence (fp32, n floats, n % 8 == 0) as bf16 for the dense products
int gru_bf16_convert_x(var_ctx* c, hipStream_t s, const float* x, long n, int maxclips, void* ws) {
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, (const float4*)x,
                       (uint4*)gru_bf16_x16(ws, maxclips), n / 8);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// n floats (n % 8 == 0) -> bf16
int gru_bf16_to_bf16(var_ctx* c, hipStream_t s, const float* x, void* y, long n) {
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, (const float4*)x, (uint4*)y, n / 8);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// once per forward: both fragment tables of W_hh, and the zero initial state's bf16 copy
int gru_bf16_pack(var_ctx* c, hipStream_t s, const float* w_hh, const float* w_ih, long dirP, int nclips, int maxclips, float* Hb,
                  long dirH, void* ws) {
    uint4* wf = (uint4*)ws;
    uint4* wb = (uint4*)((char*)ws + kWfBytes);
    GruPrep q{};
    q.w_ih = w_ih; q.wih16 = (uint4*)gru_bf16_wih16(ws, maxclips);      // the A operand of the input projection and of dX
    q.h16 = (uint4*)gru_bf16_h16(ws); q.h16dir = (long)(SEQ + 1) * nclips * GH * 2 / 16; q.nrow16 = nclips * GH * 2 / 16;
    q.hb = (uint4*)Hb; q.hbdir = dirH * 4 / 16; q.nrow32 = nclips * GH * 4 / 16;
    q.cnt = gru_sync(ws, maxclips) + 4; q.ncnt = sync_counters(maxclips);
    const int n = (int)((kWfBytes + kWbBytes) / 16) + 2 * (G3 * 448 / 8) + 2 * q.nrow16 + 2 * q.nrow32 + q.ncnt;
    hipLaunchKernelGGL(gru_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w_hh, dirP, wf, wb, q);
    VAR_HIP_CHECK(c, hipGetLastError());
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)gru_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds));
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)gru_seq_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds + 16));
        attr |= var_dev_bit(c);
    }
    return VAR_OK;
}

int gru_bf16_step_fwd(var_ctx* c, hipStream_t s, const float* GI, float* Hb, const float* b_hh, long dirP, float* R, float* Z,
                      float* Nn, float* GHN, int nclips, int maxclips, int step, long dirGI, long dirH, long dirS, int save, void* ws) {
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(NJS, (nclips + 63) / 64, 2), dim3(512), kFwdLds, s, GI, Hb, (uint2*)gru_bf16_h16(ws),
                       (const uint4*)ws, b_hh, dirP, R, Z, Nn, GHN, nclips, step, dirGI, dirH, dirS, save);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int gru_bf16_step_bwd(var_ctx* c, hipStream_t s, float* DH, const float* Hb, const float* R, const float* Z, const float* Nn,
                      const float* GHN, float* DGI, float* DGH, int nclips, int maxclips, int step, int has_next, long dirGI,
                      long dirH, long dirS, long dirDGH, void* ws) {
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(NJS, (nclips + 63) / 64, 2), dim3(512), kBwdLds, s, DH, Hb,
                       (const uint4*)((const char*)ws + kWfBytes), (uint2*)gru_bf16_dgh16(ws, maxclips),
                       (uint2*)gru_bf16_dgi16(ws, maxclips), R, Z, Nn, GHN, DGI, DGH, nclips, step, has_next, dirGI, dirH, dirS, dirDGH);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// ---- one launch per pass (gru_seq_*_kernel) ---------------------------------------------------------------------
// The sequence kernels need every workgroup of the grid resident at once (one 512-thread workgroup per CU): return 1, and
// the caller takes the per-step launches, when the grid is larger than the device.
static int seq_fits(var_ctx* c, int nclips) {
    // workgroups the device can hold at once: CUs x what the occupancy query grants the larger of the two sequence kernels
    // (registers, LDS), not just the CU count -- a kernel that stops fitting one workgroup per CU must not be launched this way
    static int slots = 0;
    if (!slots) {
        int cus = 0, f = 0, b = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, (const void*)gru_seq_fwd_kernel, 512, kFwdLds + 16) != hipSuccess) f = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)gru_seq_bwd_kernel, 512, kBwdLds + 16) != hipSuccess) b = 0;
        const int per = f < b ? f : b;
        slots = cus * (per > 1 ? 1 : per);        // one workgroup per CU is what the hand-off timing was measured for
        if (slots < 1) slots = -1;
    }
    return slots > 0 && 2 * NJS * ((nclips + 63) / 64) <= slots;
}

int gru_bf16_reset_timeout(var_ctx* c, hipStream_t s, int maxclips, void* ws) {
    { const int rz = var_zero_async(c, s, gru_sync(ws, maxclips), 16); if (rz != VAR_OK) return rz; }
    return VAR_OK;
}

int gru_bf16_seq_fwd(var_ctx* c, hipStream_t s, const float* GI, float* Hb, const float* b_hh, long dirP, float* R, float* Z, float* Nn,
                     float* GHN, int nclips, int maxclips, long dirGI, long dirH, long dirS, int save, void* ws, int drop_one) {
    if (!seq_fits(c, nclips)) return 1;
    unsigned* sync = gru_sync(ws, maxclips);                 // (its counters were zeroed by gru_bf16_pack, which precedes every forward)
    // drop_one (tests): one hidden slice of every group is not launched, so no group ever completes -- what a grid that is
    // not fully resident looks like to the others: their waits must expire and the launch must end
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(drop_one ? NJS - 1 : NJS, (nclips + 63) / 64, 2), dim3(512), kFwdLds + 16, s, GI, Hb, (uint2*)gru_bf16_h16(ws),
                       (const uint4*)ws, b_hh, dirP, R, Z, Nn, GHN, nclips, dirGI, dirH, dirS, save, sync + 4, sync);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// grads: the parameter gradient (n floats), overwritten with NaN by a trailing check if either sequence kernel timed out
int gru_bf16_seq_bwd(var_ctx* c, hipStream_t s, float* DH, const float* Hb, const float* R, const float* Z, const float* Nn,
                     const float* GHN, float* DGI, float* DGH, int nclips, int maxclips, long dirGI, long dirH, long dirS, long dirDGH,
                     void* ws, int store32) {
    if (!seq_fits(c, nclips)) return 1;
    unsigned* sync = gru_sync(ws, maxclips);
    unsigned* cnt = sync + 4 + sync_counters(maxclips);
    { const int rz = var_zero_async(c, s, cnt, 4L * sync_counters(maxclips)); if (rz != VAR_OK) return rz; }
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(NJS, (nclips + 63) / 64, 2), dim3(512), kBwdLds + 16, s, DH, Hb,
                       (const uint4*)((const char*)ws + kWfBytes), (uint2*)gru_bf16_dgh16(ws, maxclips),
                       (uint2*)gru_bf16_dgi16(ws, maxclips), R, Z, Nn, GHN, DGI, DGH, nclips, dirGI, dirH, dirS, dirDGH, cnt, sync,
                       gru_bf16_bias_part(ws, maxclips), store32);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int gru_bf16_poison_on_timeout(var_ctx* c, hipStream_t s, float* grads, int n, int maxclips, void* ws) {
    hipLaunchKernelGGL(gru_poison_kernel, dim3((n + 255) / 256), dim3(256), 0, s, gru_sync(ws, maxclips), grads, n);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// The status words (gru_sync(ws)[0..3]): [0] = code of a time-out in the CURRENT step (what Adam's guard and the poison kernel
// read), [1] = number of earlier steps that timed out, [2] = the code of the last of them.
__global__ void gru_step_begin_kernel(unsigned* __restrict__ w) {
    if (w[0]) { w[1] += 1; w[2] = w[0]; w[0] = 0; }
}

int gru_bf16_step_begin(var_ctx* c, hipStream_t s, int maxclips, void* ws) {
    hipLaunchKernelGGL(gru_step_begin_kernel, dim3(1), dim3(1), 0, s, gru_sync(ws, maxclips));
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

const unsigned* gru_bf16_timeout_ptr(int maxclips, void* ws) { return gru_sync(ws, maxclips); }

// blocking: the time-out word (0 = every hand-off of every sequence launch so far completed); a time-out of an earlier
// step reads as 0x40000000 | its code
int gru_bf16_timeout_word(var_ctx* c, int maxclips, void* ws, unsigned* out) {
    unsigned w[4] = {0, 0, 0, 0};
    VAR_HIP_CHECK(c, hipMemcpy(w, gru_sync(ws, maxclips), 16, hipMemcpyDeviceToHost));
    *out = w[0] ? w[0] : (w[1] ? (0x40000000u | w[2]) : 0u);
    return VAR_OK;
}
