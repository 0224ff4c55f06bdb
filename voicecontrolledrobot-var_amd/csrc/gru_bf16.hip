// The recurrence of the iTHOR model's bidirectional GRU(448 -> 512) in its bf16 mode (models/pretext/ai2thor_pretext_model.py:
// 31-33; torch.nn.GRU gate order r, z, n): ONE kernel per time step and pass that does the recurrent product on
// v_mfma_f32_32x32x16_bf16 AND the gate arithmetic, instead of a split-K gather-GEMM launch + its partial slabs + a gate
// kernel per step (32 us per step; the 146 dependent steps were 25 % of the bf16 training step).
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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// W_hh (dir stride dirP floats; [1536][512]) -> both fragment tables
__global__ void __launch_bounds__(256) gru_pack_kernel(const float* __restrict__ w_hh, long dirP, uint4* __restrict__ wf,
                                                       uint4* __restrict__ wb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nf = 2 * NJS * 32 * 3 * 64, nb = 2 * NJS * 96 * 64;
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
__global__ void __launch_bounds__(256) gru_step_fwd_kernel(const float* __restrict__ GI, float* __restrict__ Hb, const uint4* __restrict__ wf,
                                                           const float* __restrict__ b_hh, long dirP, float* __restrict__ R,
                                                           float* __restrict__ Z, float* __restrict__ Nn, float* __restrict__ GHN,
                                                           int nclips, int step, long dirGI, long dirH, long dirS, int save) {
    __shared__ float red[2][3][16][64];                  // the upper K half's accumulators
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, cbk = wave & 1, kh = wave >> 1;
    const int js = blockIdx.x, cs = blockIdx.y, dir = blockIdx.z;
    const int clip = 64 * cs + 32 * cbk + (lane & 31);
    const bool live = clip < nclips;
    const float* hprev = Hb + dir * dirH + (long)step * nclips * GH + (long)(live ? clip : 0) * GH;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, (int)kWfBytes, 0x00020000);
    const int wbase = ((dir * NJS + js) * 32 + 16 * kh) * 3 * 1024;
    f32x16_t acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    constexpr int CH = 4;                                // k-steps per chunk, two chunks in flight
    u32x4_t a[2][CH][3];
    float4 b[2][CH][2];
    auto load = [&](int c, int set) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ks = CH * c + i;
#pragma unroll
            for (int g = 0; g < 3; ++g) a[set][i][g] = wload(wr, lane * 16, wbase + (ks * 3 + g) * 1024);
            const float4* src = (const float4*)(hprev + 16 * (16 * kh + ks) + 8 * h);
            b[set][i][0] = src[0]; b[set][i][1] = src[1];
        }
    };
    load(0, 0);
#pragma unroll
    for (int c = 0; c < 16 / CH; ++c) {
        const int set = c & 1;
        if (c + 1 < 16 / CH) load(c + 1, set ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bf16x8_t bv = to_bf16x8(b[set][i][0], b[set][i][1]);
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[set][i][g]), bv, acc[g], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kh == 1) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[cbk][g][r][lane] = acc[g][r];
    }
    __syncthreads();
    if (kh == 1 || !live) return;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] += red[cbk][g][r][lane];
    // gates on the accumulator layout: register quad q holds hidden units 32 js + 8 q + 4 h + 0..3 of this lane's clip
    const int t = dir ? SEQ - 1 - step : step;
    const float* gi = GI + dir * dirGI + ((long)clip * SEQ + t) * G3;
    const float* bh = b_hh + dir * dirP;
    float* hnext = Hb + dir * dirH + (long)(step + 1) * nclips * GH + (long)clip * GH;
    const long so = dir * dirS + (long)step * nclips * GH + (long)clip * GH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 32 * js + 8 * q + 4 * h;
        const float4 gr = *(const float4*)(gi + j), gz = *(const float4*)(gi + GH + j), gn = *(const float4*)(gi + 2 * GH + j);
        const float4 br = *(const float4*)(bh + j), bz = *(const float4*)(bh + GH + j), bn = *(const float4*)(bh + 2 * GH + j);
        const float4 hp = *(const float4*)(hprev + j);
        float4 o, rr, zz, nn, gg;
#define GRU_LANE(c, e)                                                         \
        {                                                                      \
            const float r_ = sigmoidf_(gr.c + (acc[0][4 * q + e] + br.c));     \
            const float z_ = sigmoidf_(gz.c + (acc[1][4 * q + e] + bz.c));     \
            const float ghn_ = acc[2][4 * q + e] + bn.c;                       \
            const float n_ = tanhf(gn.c + r_ * ghn_);                          \
            o.c = (1.f - z_) * n_ + z_ * hp.c;                                 \
            rr.c = r_; zz.c = z_; nn.c = n_; gg.c = ghn_;                      \
        }
        GRU_LANE(x, 0) GRU_LANE(y, 1) GRU_LANE(z, 2) GRU_LANE(w, 3)
#undef GRU_LANE
        *(float4*)(hnext + j) = o;
        if (save) {
            *(float4*)(R + so + j) = rr; *(float4*)(Z + so + j) = zz; *(float4*)(Nn + so + j) = nn; *(float4*)(GHN + so + j) = gg;
        }
    }
}

// ---- backward step ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gru_step_bwd_kernel(float* __restrict__ DH, const float* __restrict__ Hb, const uint4* __restrict__ wb,
                                                           const float* __restrict__ R, const float* __restrict__ Z,
                                                           const float* __restrict__ Nn, const float* __restrict__ GHN,
                                                           float* __restrict__ DGI, float* __restrict__ DGH, int nclips, int step,
                                                           int has_next, long dirGI, long dirH, long dirS, long dirDGH) {
    __shared__ float red[2][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, cbk = wave & 1, kh = wave >> 1;
    const int js = blockIdx.x, cs = blockIdx.y, dir = blockIdx.z;
    const int clip = 64 * cs + 32 * cbk + (lane & 31);
    const bool live = clip < nclips;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (has_next) {                                      // uniform
        const float* dgn = DGH + dir * dirDGH + ((long)(step + 1) * nclips + (live ? clip : 0)) * G3;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)kWbBytes, 0x00020000);
        const int wbase = ((dir * NJS + js) * 96 + 48 * kh) * 1024;
        constexpr int CH = 8;
        u32x4_t a[2][CH];
        float4 b[2][CH][2];
        auto load = [&](int c, int set) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ks = CH * c + i;
                a[set][i] = wload(wr, lane * 16, wbase + ks * 1024);
                const float4* src = (const float4*)(dgn + 16 * (48 * kh + ks) + 8 * h);
                b[set][i][0] = src[0]; b[set][i][1] = src[1];
            }
        };
        load(0, 0);
#pragma unroll
        for (int c = 0; c < 48 / CH; ++c) {
            const int set = c & 1;
            if (c + 1 < 48 / CH) load(c + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[set][i]),
                                                              to_bf16x8(b[set][i][0], b[set][i][1]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[cbk][r][lane] = acc[r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[cbk][r][lane];
        }
    }
    if (kh == 1 || !live) return;
    const int t = dir ? SEQ - 1 - step : step;
    const long so = dir * dirS + (long)step * nclips * GH + (long)clip * GH;
    const float* hprev = Hb + dir * dirH + (long)step * nclips * GH + (long)clip * GH;
    float* dh = DH + (long)dir * nclips * GH + (long)clip * GH;
    float* dgi = DGI + dir * dirGI + ((long)clip * SEQ + t) * G3;
    float* dgh = DGH + dir * dirDGH + ((long)step * nclips + clip) * G3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 32 * js + 8 * q + 4 * h;
        const float4 r4 = *(const float4*)(R + so + j), z4 = *(const float4*)(Z + so + j), n4 = *(const float4*)(Nn + so + j);
        const float4 g4 = *(const float4*)(GHN + so + j), hp = *(const float4*)(hprev + j), d4 = *(const float4*)(dh + j);
        float4 dr, dz, dn, dnr, dd;
#define GRU_LANE(c, e)                                                         \
        {                                                                      \
            const float dh_ = d4.c + acc[4 * q + e];                           \
            const float dn_ = dh_ * (1.f - z4.c) * (1.f - n4.c * n4.c);        \
            dz.c = dh_ * (hp.c - n4.c) * z4.c * (1.f - z4.c);                  \
            dr.c = dn_ * g4.c * r4.c * (1.f - r4.c);                           \
            dn.c = dn_; dnr.c = dn_ * r4.c; dd.c = dh_ * z4.c;                 \
        }
        GRU_LANE(x, 0) GRU_LANE(y, 1) GRU_LANE(z, 2) GRU_LANE(w, 3)
#undef GRU_LANE
        *(float4*)(dgi + j) = dr; *(float4*)(dgi + GH + j) = dz; *(float4*)(dgi + 2 * GH + j) = dn;
        *(float4*)(dgh + j) = dr; *(float4*)(dgh + GH + j) = dz; *(float4*)(dgh + 2 * GH + j) = dnr;
        *(float4*)(dh + j) = dd;
    }
}

}  // namespace

long gru_bf16_workspace_bytes() { return kWfBytes + kWbBytes; }

int gru_bf16_pack(var_ctx* c, hipStream_t s, const float* w_hh, long dirP, void* ws) {
    uint4* wf = (uint4*)ws;
    uint4* wb = (uint4*)((char*)ws + kWfBytes);
    const int n = (int)((kWfBytes + kWbBytes) / 16);
    hipLaunchKernelGGL(gru_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w_hh, dirP, wf, wb);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int gru_bf16_step_fwd(var_ctx* c, hipStream_t s, const float* GI, float* Hb, const float* b_hh, long dirP, float* R, float* Z,
                      float* Nn, float* GHN, int nclips, int step, long dirGI, long dirH, long dirS, int save, void* ws) {
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(NJS, (nclips + 63) / 64, 2), dim3(256), 0, s, GI, Hb, (const uint4*)ws, b_hh, dirP, R,
                       Z, Nn, GHN, nclips, step, dirGI, dirH, dirS, save);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int gru_bf16_step_bwd(var_ctx* c, hipStream_t s, float* DH, const float* Hb, const float* R, const float* Z, const float* Nn,
                      const float* GHN, float* DGI, float* DGH, int nclips, int step, int has_next, long dirGI, long dirH,
                      long dirS, long dirDGH, void* ws) {
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(NJS, (nclips + 63) / 64, 2), dim3(256), 0, s, DH, Hb,
                       (const uint4*)((const char*)ws + kWfBytes), R, Z, Nn, GHN, DGI, DGH, nclips, step, has_next, dirGI, dirH,
                       dirS, dirDGH);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
