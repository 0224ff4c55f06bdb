// Sound CNN backward (autograd of models/pretext/arm_pretext_model.py:21-34).
//   snd_dgrad : per clip, back-propagates d(loss)/d(flattened features) (already masked by the
//               heads kernel) through the three (3,1)/s2 convs, applying the ReLU masks; the
//               whole clip's gradients live in LDS.
//   snd_wgrad : weight/bias gradients.  A workgroup walks its share of the clips with the
//               per-thread accumulators in registers and writes one partial slab laid out
//               exactly like the soundCNN slice of the parameter arena.
//   snd_reduce: fixed-order slab sum (bitwise reproducible, no float atomics).
#include "var_common.h"

namespace {
constexpr int T0 = 100, F = 40, T1 = 48, T2 = 23, T3 = 11, T4 = 5;
constexpr int SND_SLICE = 32 * 200 + 32 + 3 * (32 * 96 + 32);   // 15744 floats: soundCNN.{0,2,4,6}.{weight,bias}

// gx[c][tin] = (sum_{n,kt : tin = 2t+kt} g[n][t] * W[n][c][kt]) * (y_in > 0)
template <int TIN, int TOUT>
__device__ __forceinline__ void dgrad1d(const float* __restrict__ g /*LDS [32][TOUT]*/,
                                        float* __restrict__ gx /*LDS [32][TIN]*/,
                                        const float* __restrict__ wd /*[3][32 n][32 c]*/,
                                        const float* __restrict__ yin /*global (32,TIN)*/,
                                        float* __restrict__ gout /*global (32,TIN)*/, int c, int grp) {
    for (int tin = grp; tin < TIN; tin += 8) {
        float s = 0.f;
        if (tin & 1) {
            const int t = (tin - 1) >> 1;
            if (t < TOUT)
                for (int n = 0; n < 32; ++n) s += g[n * TOUT + t] * wd[(1 * 32 + n) * 32 + c];
        } else {
            const int ta = tin >> 1, tb = (tin >> 1) - 1;
            for (int n = 0; n < 32; ++n) {
                if (ta < TOUT) s += g[n * TOUT + ta] * wd[(0 * 32 + n) * 32 + c];
                if (tb >= 0 && tb < TOUT) s += g[n * TOUT + tb] * wd[(2 * 32 + n) * 32 + c];
            }
        }
        const float v = yin[c * TIN + tin] > 0.f ? s : 0.f;
        gx[c * TIN + tin] = v;
        gout[c * TIN + tin] = v;
    }
}

__global__ void __launch_bounds__(256)
snd_dgrad_kernel(int clip_lo, const float* __restrict__ wd1, const float* __restrict__ wd2,
                 const float* __restrict__ wd3, const float* __restrict__ a1, const float* __restrict__ a2,
                 const float* __restrict__ a3, const float* __restrict__ g4, float* __restrict__ g3,
                 float* __restrict__ g2, float* __restrict__ g1) {
    __shared__ float s4[32 * T4], s3[32 * T3], s2[32 * T2], s1[32 * T1];
    const int clip = clip_lo + blockIdx.x;
    const int tid = threadIdx.x, c = tid & 31, grp = tid >> 5;
    if (tid < 32 * T4) s4[tid] = g4[(size_t)clip * 32 * T4 + tid];
    __syncthreads();
    dgrad1d<T3, T4>(s4, s3, wd3, a3 + (size_t)clip * 32 * T3, g3 + (size_t)clip * 32 * T3, c, grp);
    __syncthreads();
    dgrad1d<T2, T3>(s3, s2, wd2, a2 + (size_t)clip * 32 * T2, g2 + (size_t)clip * 32 * T2, c, grp);
    __syncthreads();
    dgrad1d<T1, T2>(s2, s1, wd1, a1 + (size_t)clip * 32 * T1, g1 + (size_t)clip * 32 * T1, c, grp);
}

template <int TIN, int TOUT>
__device__ __forceinline__ void wgrad1d(float (&acc)[4][3], float& bacc, const float* __restrict__ g /*[32][TOUT|1]*/,
                                        const float* __restrict__ in /*[32][TIN]*/, int n, int cg, bool do_bias) {
    constexpr int GS = TOUT | 1;
    for (int t = 0; t < TOUT; ++t) {
        const float gv = g[n * GS + t];
        if (do_bias) bacc += gv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cg + 8 * i;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) acc[i][kt] += gv * in[c * TIN + 2 * t + kt];
        }
    }
}

template <int TT>
__device__ __forceinline__ void stage_padded(float* dst, const float* __restrict__ src, int tid) {
    constexpr int GS = TT | 1;
    for (int e = tid; e < 32 * TT; e += 256) dst[(e / TT) * GS + (e % TT)] = src[e];
}

__global__ void __launch_bounds__(256)
snd_wgrad_kernel(int clip_lo, int clip_hi, int B, const float* __restrict__ pos, const float* __restrict__ neg,
                 const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ a3,
                 const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ g3,
                 const float* __restrict__ g4, float* __restrict__ slabs) {
    __shared__ float x[T0 * F], y1[32 * T1], y2[32 * T2], y3[32 * T3];
    __shared__ float q1[32 * (T1 | 1)], q2[32 * (T2 | 1)], q3[32 * (T3 | 1)], q4[32 * (T4 | 1)];
    const int tid = threadIdx.x, n = tid & 31, cg = tid >> 5;
    float acc0[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) acc0[i] = 0.f;
    float acc1[4][3], acc2[4][3], acc3[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) { acc1[i][k] = 0.f; acc2[i][k] = 0.f; acc3[i][k] = 0.f; }
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
    const bool do_bias = cg == 0;

    for (int clip = clip_lo + blockIdx.x; clip < clip_hi; clip += gridDim.x) {
        const float* src = clip < B ? pos + (size_t)clip * T0 * F : neg + (size_t)(clip - B) * T0 * F;
        __syncthreads();
        for (int e = tid; e < T0 * F / 4; e += 256) ((float4*)x)[e] = ((const float4*)src)[e];
        for (int e = tid; e < 32 * T1; e += 256) y1[e] = a1[(size_t)clip * 32 * T1 + e];
        for (int e = tid; e < 32 * T2; e += 256) y2[e] = a2[(size_t)clip * 32 * T2 + e];
        for (int e = tid; e < 32 * T3; e += 256) y3[e] = a3[(size_t)clip * 32 * T3 + e];
        stage_padded<T1>(q1, g1 + (size_t)clip * 32 * T1, tid);
        stage_padded<T2>(q2, g2 + (size_t)clip * 32 * T2, tid);
        stage_padded<T3>(q3, g3 + (size_t)clip * 32 * T3, tid);
        stage_padded<T4>(q4, g4 + (size_t)clip * 32 * T4, tid);
        __syncthreads();
        // conv0: dW0[n][k] = sum_t g1[n][t] * x[80 t + k],  k = cg + 8 i
        for (int t = 0; t < T1; ++t) {
            const float gv = q1[n * (T1 | 1) + t];
            if (do_bias) bs0 += gv;
#pragma unroll
            for (int i = 0; i < 25; ++i) acc0[i] += gv * x[80 * t + cg + 8 * i];
        }
        wgrad1d<T1, T2>(acc1, bs1, q2, y1, n, cg, do_bias);
        wgrad1d<T2, T3>(acc2, bs2, q3, y2, n, cg, do_bias);
        wgrad1d<T3, T4>(acc3, bs3, q4, y3, n, cg, do_bias);
    }
    float* slab = slabs + (size_t)blockIdx.x * SND_SLICE;
#pragma unroll
    for (int i = 0; i < 25; ++i) slab[n * 200 + cg + 8 * i] = acc0[i];
    if (do_bias) slab[6400 + n] = bs0;
    float* sl = slab + 6432;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int o = (n * 32 + cg + 8 * i) * 3 + kt;
            sl[o] = acc1[i][kt];
            sl[3104 + o] = acc2[i][kt];
            sl[6208 + o] = acc3[i][kt];
        }
    if (do_bias) { sl[3072 + n] = bs1; sl[3104 + 3072 + n] = bs2; sl[6208 + 3072 + n] = bs3; }
}

__global__ void __launch_bounds__(256)
snd_reduce_kernel(const float* __restrict__ slabs, int G, float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SND_SLICE) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += slabs[(size_t)g * SND_SLICE + e];
    out[e] = s;
}
}  // namespace

static const int kSndG = 64;
size_t snd_slab_floats() { return (size_t)kSndG * SND_SLICE; }

// clips [lo,hi) of the (pos | neg) stack took part in the forward
int launch_snd_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    const int lo = c->saved_pos ? 0 : B, hi = c->saved_neg ? 2 * B : B;
    if (hi <= lo) {
        VAR_HIP_CHECK(c, hipMemsetAsync(grads + L.snd_w[0], 0, sizeof(float) * SND_SLICE, s));
        return VAR_OK;
    }
    { ProfScope prof(c, s, TAG_SND_DGRAD);
    hipLaunchKernelGGL(snd_dgrad_kernel, dim3(hi - lo), dim3(256), 0, s, lo,
                       c->wpack + K.snd_d[1], c->wpack + K.snd_d[2], c->wpack + K.snd_d[3],
                       c->sact[1], c->sact[2], c->sact[3], c->gsact[4], c->gsact[3], c->gsact[2], c->gsact[1]); }
    int G = hi - lo < kSndG ? hi - lo : kSndG;
    float* slabs = c->slabs + c->snd_slab_off;
    { ProfScope prof(c, s, TAG_SND_WGRAD);
    hipLaunchKernelGGL(snd_wgrad_kernel, dim3(G), dim3(256), 0, s, lo, hi, B, c->saved_pos, c->saved_neg,
                       c->sact[1], c->sact[2], c->sact[3], c->gsact[1], c->gsact[2], c->gsact[3], c->gsact[4], slabs); }
    ProfScope prof(c, s, TAG_SND_REDUCE);
    hipLaunchKernelGGL(snd_reduce_kernel, dim3((SND_SLICE + 255) / 256), dim3(256), 0, s, slabs, G,
                       grads + L.snd_w[0]);
    VAR_HIP_CHECK(c, hipGetLastError());
    (void)params;
    return VAR_OK;
}
