// In-batch-negatives contrastive head (BASELINE.json north_star / configs[2]; an EXTENSION -- the reference trains
// with the explicit-negative triplet loss only): every anchor image embedding a_i is scored
// against ALL candidate sound embeddings c_j of the global batch (the positives and the explicit negatives of every
// sample, all-gathered over the ranks) with a softmax over negative pairwise distances
//     d_ij = || a_i - c_j + 1e-6 ||_2   (torch's pairwise_distance convention, as in TripletMarginLoss)
//     L = inv_count * sum_i [ logsumexp_j(-d_ij / tau) + d_{i,t(i)} / tau ],   t(i) = column of sample i's positive.
// One wavefront per anchor row: lanes stride over the candidates, online max/sum per lane, wave-shuffle combine
// (no LDS, no atomics); a second kernel, one wavefront per candidate, forms the candidate gradients from the saved
// row log-sum-exps (so each rank's partial candidate gradient is bitwise reproducible; under data parallelism the
// partials are summed with one small all-reduce and each rank keeps the slice of its own clips).
#include <math.h>

#include "var_common.h"

namespace {
constexpr float kPdEps = 1e-6f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float dist3(const float* a, const float* c, float& ux, float& uy, float& uz) {
    ux = a[0] - c[0] + kPdEps; uy = a[1] - c[1] + kPdEps; uz = a[2] - c[2] + kPdEps;
    return sqrtf(ux * ux + uy * uy + uz * uz);
}

// rows: lse_i, row loss, gradient wrt the anchor
__global__ void __launch_bounds__(256) inbatch_rows_kernel(const float* __restrict__ a, const float* __restrict__ cand,
                                                          const int* __restrict__ target, int B, int M, float inv_tau,
                                                          float inv_count, float* __restrict__ lse, float* __restrict__ rowloss,
                                                          float* __restrict__ ga) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= B) return;
    const float* ai = a + 3 * i;
    float mx = -INFINITY, sm = 0.f;
    for (int j = lane; j < M; j += 64) {
        float ux, uy, uz;
        const float l = -dist3(ai, cand + 3 * j, ux, uy, uz) * inv_tau;
        const float nm = fmaxf(mx, l);
        sm = sm * expf(mx - nm) + expf(l - nm);
        mx = nm;
    }
    const float gmx = wave_max(mx);
    const float gs = wave_sum(sm * expf(mx - gmx));          // lanes without a candidate hold sm = 0, mx = -inf -> 0
    const float L = gmx + logf(gs);
    const int t = target[i];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int j = lane; j < M; j += 64) {
        float ux, uy, uz;
        const float d = dist3(ai, cand + 3 * j, ux, uy, uz);
        const float s = expf(-d * inv_tau - L) - (j == t ? 1.f : 0.f);
        const float w = -s * inv_tau * inv_count / fmaxf(d, 1e-12f);      // dL/dd_ij / d_ij
        gx += w * ux; gy += w * uy; gz += w * uz;
    }
    gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz);
    if (lane == 0) {
        float ux, uy, uz;
        const float dt = dist3(ai, cand + 3 * t, ux, uy, uz);
        lse[i] = L;
        rowloss[i] = (L + dt * inv_tau) * inv_count;
        ga[3 * i] = gx; ga[3 * i + 1] = gy; ga[3 * i + 2] = gz;
    }
}

// columns: gradient wrt candidate j from the B local rows (partial under data parallelism)
// (the block after the last column block adds the row losses up in a fixed order: the sum only needs the rows kernel, and a
//  launch of its own cost 6 us on the step's critical path)
__device__ __forceinline__ void inbatch_loss_sum(const float* __restrict__ rowloss, int B, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) acc += rowloss[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void __launch_bounds__(256) inbatch_cols_kernel(const float* __restrict__ a, const float* __restrict__ cand,
                                                          const int* __restrict__ target, const float* __restrict__ lse, int B,
                                                          int M, float inv_tau, float inv_count, float* __restrict__ gc,
                                                          const float* __restrict__ rowloss, float* __restrict__ loss_out) {
    if ((int)blockIdx.x == (M + 3) / 4) { inbatch_loss_sum(rowloss, B, loss_out); return; }
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= M) return;
    const float* cj = cand + 3 * j;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int i = lane; i < B; i += 64) {
        float ux, uy, uz;
        const float d = dist3(a + 3 * i, cj, ux, uy, uz);
        const float s = expf(-d * inv_tau - lse[i]) - (target[i] == j ? 1.f : 0.f);
        const float w = -s * inv_tau * inv_count / fmaxf(d, 1e-12f);
        gx -= w * ux; gy -= w * uy; gz -= w * uz;
    }
    gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz);
    if (lane == 0) { gc[3 * j] = gx; gc[3 * j + 1] = gy; gc[3 * j + 2] = gz; }
}

}  // namespace

extern "C" int var_inbatch_loss_fwd_bwd(var_ctx* c, void* stream, const float* anchor, const float* cand, const int* target,
                                        int B, int M, float tau, float inv_count, float* scratch, float* loss_out,
                                        float* g_anchor, float* g_cand) {
    if (!c) return VAR_ERR_ARG;
    if (!anchor || !cand || !target || !scratch || !loss_out || !g_anchor || !g_cand || B < 1 || M < 1 || !(tau > 0.f)) {
        VAR_SET_ERR(c, "var_inbatch_loss_fwd_bwd: bad argument");
        return VAR_ERR_ARG;
    }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    float* lse = scratch;            // B
    float* rowloss = scratch + B;    // B
    hipLaunchKernelGGL(inbatch_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, s, anchor, cand, target, B, M, 1.f / tau, inv_count,
                       lse, rowloss, g_anchor);
    VAR_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(inbatch_cols_kernel, dim3((M + 3) / 4 + 1), dim3(256), 0, s, anchor, cand, target, lse, B, M, 1.f / tau,
                       inv_count, g_cand, rowloss, loss_out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
