// Sound CNN forward (models/pretext/arm_pretext_model.py:21-34):
//   Conv2d(1,32,(5,40),s(2,1))+ReLU -> 3 x [Conv2d(32,32,(3,1),s(2,1))+ReLU] -> Flatten (c*5+t)
// on (clip,1,100,40) MFCC.  The 5x40 kernel spans the full feature width, so every layer is a
// 1-D convolution over time; one workgroup keeps a whole clip and all four activations in LDS
// (27 KB) and runs the four layers back to back -- the clip is read from HBM exactly once.
// Thread (n = tid&31, g = tid>>5) produces channel n at times g, g+8, ...; the filter value
// W[k][n] (packed image, L2-resident) is loaded once per k and reused for all its times.
#include "var_common.h"

namespace {
constexpr int T0 = 100, F = 40, T1 = 48, T2 = 23, T3 = 11, T4 = 5;

template <int TIN, int TOUT, int NT>
__device__ __forceinline__ void conv1d_k3(const float* __restrict__ in /*LDS [32][TIN]*/,
                                          float* __restrict__ out /*LDS [32][TOUT]*/,
                                          const float* __restrict__ w /*[96][32]*/,
                                          const float* __restrict__ bias, float* __restrict__ gout,
                                          int n, int g) {
    float acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float wv = w[(kt * 32 + c) * 32 + n];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int t = g + 8 * j;
                if (t < TOUT) acc[j] += wv * in[c * TIN + 2 * t + kt];
            }
        }
    }
    const float bv = bias[n];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = g + 8 * j;
        if (t < TOUT) {
            float v = acc[j] + bv;
            v = v > 0.f ? v : 0.f;
            out[n * TOUT + t] = v;
            gout[n * TOUT + t] = v;
        }
    }
}
}  // namespace

__global__ void __launch_bounds__(256)
snd_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg, int B,
               const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
               const float* __restrict__ w3, const float* __restrict__ b0, const float* __restrict__ b1,
               const float* __restrict__ b2, const float* __restrict__ b3,
               float* __restrict__ a1, float* __restrict__ a2, float* __restrict__ a3, float* __restrict__ a4) {
    __shared__ float x[T0 * F];
    __shared__ float y1[32 * T1], y2[32 * T2], y3[32 * T3], y4[32 * T4];
    const int clip = blockIdx.x;              // [0,B) positive, [B,2B) negative
    const float* src = clip < B ? (pos ? pos + (size_t)clip * T0 * F : nullptr)
                                : (neg ? neg + (size_t)(clip - B) * T0 * F : nullptr);
    if (!src) return;                         // branch absent for this call (uniform per block)
    const int tid = threadIdx.x, n = tid & 31, g = tid >> 5;
    for (int e = tid; e < T0 * F / 4; e += 256) ((float4*)x)[e] = ((const float4*)src)[e];
    __syncthreads();
    {   // conv (5,40) stride (2,1): window of output t = x[80 t .. 80 t + 199]
        float acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = 0.f;
#pragma unroll 8
        for (int k = 0; k < 200; ++k) {
            const float wv = w0[k * 32 + n];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] += wv * x[80 * (g + 8 * j) + k];
        }
        const float bv = b0[n];
        float* go = a1 + (size_t)clip * 32 * T1;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int t = g + 8 * j;
            float v = acc[j] + bv;
            v = v > 0.f ? v : 0.f;
            y1[n * T1 + t] = v;
            go[n * T1 + t] = v;
        }
    }
    __syncthreads();
    conv1d_k3<T1, T2, 3>(y1, y2, w1, b1, a2 + (size_t)clip * 32 * T2, n, g);
    __syncthreads();
    conv1d_k3<T2, T3, 2>(y2, y3, w2, b2, a3 + (size_t)clip * 32 * T3, n, g);
    __syncthreads();
    conv1d_k3<T3, T4, 1>(y3, y4, w3, b3, a4 + (size_t)clip * 32 * T4, n, g);
}

int launch_snd_fwd(var_ctx* c, hipStream_t s, const float* params, const float* pos, const float* neg, int B) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    ProfScope prof(c, s, TAG_SND_FWD);
    hipLaunchKernelGGL(snd_fwd_kernel, dim3(2 * B), dim3(256), 0, s, pos, neg, B,
                       c->wpack + K.snd_f[0], c->wpack + K.snd_f[1], c->wpack + K.snd_f[2], c->wpack + K.snd_f[3],
                       params + L.snd_b[0], params + L.snd_b[1], params + L.snd_b[2], params + L.snd_b[3],
                       c->sact[1], c->sact[2], c->sact[3], c->sact[4]);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
