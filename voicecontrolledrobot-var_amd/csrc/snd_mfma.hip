// Sound CNN (models/pretext/arm_pretext_model.py:21-34) forward and backward on the f32 matrix cores.
//   Conv2d(1,32,(5,40),s(2,1))+ReLU -> 3 x [Conv2d(32,32,(3,1),s(2,1))+ReLU] -> Flatten (c*5+t)
// The (5,40) kernel spans the whole feature axis, so every layer is a 1-D convolution over time and
// every layer is a small GEMM  D[n][pixel] = sum_k W[n][k] * X[k][pixel]  with pixel = (clip, t):
//   layer 0: k = kt*40 + f,  X = x[clip][2t+kt][f]      (K = 200)
//   layer l: k = kt*32 + c,  X = y_l[clip][c][2t+kt]     (K = 96)
// fwd   : one workgroup keeps NC clips and all their activations in LDS and runs the four layers back
//         to back; filter rows come from the packed images in L2, prefetched a block ahead in registers.
// dgrad : per NC clips, gx[c][tin] = sum_{kt,n} Wd[kt][n][c] * g[n][(tin-kt)/2], split by the parity of
//         tin (even: kt in {0,2}; odd: kt = 1) so that each class is dense; ReLU masks applied.
// wgrad : D[n][c] per tap with K = (clip, t); 9 waves own fixed tile sets and keep their accumulators
//         in registers while the workgroup walks its clips; one partial slab per workgroup, then a
//         fixed-order reduction (bitwise reproducible, no float atomics).
#include "var_common.h"

namespace {
constexpr int T0 = 100, F = 40, T1 = 48, T2 = 23, T3 = 11, T4 = 5;
constexpr int XROW = 41;                         // padded MFCC row (bank spread for stride-2 windows)
constexpr int XCLIP = T0 * XROW;                 // 4100
constexpr int SND_SLICE = 32 * 200 + 32 + 3 * (32 * 96 + 32);   // 15744 floats: soundCNN.{0,2,4,6}.{weight,bias}

template <int NT>
__device__ __forceinline__ void zero_lds(float* p, int n, int tid) {
    for (int e = tid; e < n; e += NT) p[e] = 0.f;
}

// stage one clip (100 x 40 contiguous floats) into rows of XROW
template <int NT>
__device__ __forceinline__ void stage_clip(float* dst, const float* __restrict__ src, bool valid, int tid) {
    constexpr int TOT = T0 * F / 4;              // 1000 float4
#pragma unroll 1
    for (int e0 = tid; e0 < TOT; e0 += NT * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * NT;
            v[u] = (valid && e < TOT) ? ((const float4*)src)[e] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * NT;
            if (e < TOT) {
                const int r = e / 10, f4 = e - r * 10;
                float* d = dst + r * XROW + f4 * 4;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// 4 clips and 6 waves per workgroup (round 3; was 2 and 3): 128 fat workgroups keep fewer CUs away from the one-workgroup-
// per-CU image kernels they run beside than 256 thin ones -- step 0.3061-0.3082 -> 0.3004-0.3023 ms (1 clip: 0.311; 5: no better)
constexpr int FNC = 4, FNW = 6, FNT = FNW * 64;
static_assert(FNC * 48 == FNW * 32, "layer 0: one 32-pixel block of the FNC x 48 (clip, t) pixels per wave");
constexpr int P1 = 49, P2 = 23, P3 = 11;         // LDS row strides of y1,y2,y3 ([c][t])
constexpr int F_XS = 0, F_Y1 = FNC * XCLIP, F_Y2 = F_Y1 + FNC * 32 * P1, F_Y3 = F_Y2 + FNC * 32 * P2,
              F_END = F_Y3 + FNC * 32 * P3;

// one 1-D conv layer (k3 s2, 32->32) for the NC clips in LDS; pixel blocks dealt to waves
template <int TIN, int TOUT, int PIN, int POUT, bool LAST>
__device__ __forceinline__ void snd_layer(const float* __restrict__ in, float* __restrict__ out,
                                          const float* __restrict__ w /*[96][32]*/, const float* __restrict__ bias,
                                          float* __restrict__ gout, int clip0, int nclips, int wave, int lane) {
    constexpr int NPIX = FNC * TOUT, NPB = (NPIX + 31) / 32;
    if (wave >= NPB) return;
    const int half = lane >> 5, l31 = lane & 31;
    int p = wave * 32 + l31;
    const bool pv = p < NPIX;
    if (!pv) p = 0;
    const int cl = p / TOUT, t = p - cl * TOUT;
    const float* bp = in + cl * 32 * PIN + half * PIN + 2 * t;
    const float* wl = w + half * 32 + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float wb[3][16];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int u = 0; u < 16; ++u) wb[kt][u] = wl[(kt * 32 + 2 * u) * 32];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kt][u], bp[2 * u * PIN + kt], acc, 0, 0, 0);
    if (pv && cl < nclips) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[r] + bias[n];
            v = v > 0.f ? v : 0.f;
            if constexpr (!LAST) out[cl * 32 * POUT + n * POUT + t] = v;
            gout[(size_t)(clip0 + cl) * 32 * TOUT + n * TOUT + t] = v;
        }
    }
}

__global__ void __launch_bounds__(FNT)
snd_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg, int B, int clip_lo, int clip_hi,
               const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
               const float* __restrict__ w3, const float* __restrict__ b0, const float* __restrict__ b1,
               const float* __restrict__ b2, const float* __restrict__ b3,
               float* __restrict__ a1, float* __restrict__ a2, float* __restrict__ a3, float* __restrict__ a4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int clip0 = clip_lo + blockIdx.x * FNC;
    const int nclips = (clip_hi - clip0) < FNC ? (clip_hi - clip0) : FNC;
#pragma unroll 1
    for (int c = 0; c < FNC; ++c) {
        const int clip = clip0 + c;
        const bool valid = c < nclips;
        const float* src = clip < B ? pos + (size_t)clip * T0 * F : neg + (size_t)(clip - B) * T0 * F;
        stage_clip<FNT>(lds + F_XS + c * XCLIP, valid ? src : pos, valid, tid);
    }
    __syncthreads();
    {   // layer 0: FNC x 48 (clip,t) pixels = FNW blocks of 32, one per wave; K = 200 = 5 blocks of 20 k-pairs
        const int p = wave * 32 + l31;
        const int cl = p / T1, t = p - cl * T1;
        const float* bp = lds + F_XS + cl * XCLIP + 2 * t * XROW + half;
        const float* wl = w0 + half * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float wb[2][20];
#pragma unroll
        for (int u = 0; u < 20; ++u) wb[0][u] = wl[(2 * u) * 32];
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
            if (kt + 1 < 5) {
#pragma unroll
                for (int u = 0; u < 20; ++u) wb[(kt + 1) & 1][u] = wl[((kt + 1) * 40 + 2 * u) * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 20; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kt & 1][u], bp[kt * XROW + 2 * u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cl < nclips) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[r] + b0[n];
                v = v > 0.f ? v : 0.f;
                lds[F_Y1 + cl * 32 * P1 + n * P1 + t] = v;
                a1[(size_t)(clip0 + cl) * 32 * T1 + n * T1 + t] = v;
            }
        }
    }
    __syncthreads();
    snd_layer<T1, T2, P1, P2, false>(lds + F_Y1, lds + F_Y2, w1, b1, a2, clip0, nclips, wave, lane);
    __syncthreads();
    snd_layer<T2, T3, P2, P3, false>(lds + F_Y2, lds + F_Y3, w2, b2, a3, clip0, nclips, wave, lane);
    __syncthreads();
    snd_layer<T3, T4, P3, 1, true>(lds + F_Y3, nullptr, w3, b3, a4, clip0, nclips, wave, lane);
}

// ------------------------------------------------------------------------------------------
// dgrad: g4 -> g3 -> g2 -> g1 (each masked by its activation > 0)
// ------------------------------------------------------------------------------------------
constexpr int DNC = 4, DNW = 4, DNT = DNW * 64;
// gradient rows in LDS: [n][TOUT + 2] with a zero at both ends (t = -1 and t = TOUT)
constexpr int G4S = T4 + 2, G3S = T3 + 2, G2S = T2 + 2;
constexpr int D_G4 = 0, D_G3 = D_G4 + DNC * 32 * G4S, D_G2 = D_G3 + DNC * 32 * G3S, D_END = D_G2 + DNC * 32 * G2S;

// gx[c][tin] for tin of parity PAR; pixels = (clip, j), tin = 2j + PAR
template <int TIN, int TOUT, int GS, int GXS, bool TO_LDS>
__device__ __forceinline__ void snd_dgrad_layer(const float* __restrict__ g /*LDS [cl][n][GS]*/, float* __restrict__ gx_lds,
                                                const float* __restrict__ wd /*[3][n][c]*/,
                                                const float* __restrict__ yin, float* __restrict__ gout,
                                                int clip0, int nclips, int wave, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int NE = (TIN + 1) / 2, NO = TIN / 2;             // even / odd positions per clip
    constexpr int PBE = (DNC * NE + 31) / 32, PBO = (DNC * NO + 31) / 32;
#pragma unroll 1
    for (int it = wave; it < PBE + PBO; it += DNW) {
        const bool odd = it >= PBE;
        const int pb = odd ? it - PBE : it;
        const int per = odd ? NO : NE;
        int p = pb * 32 + l31;
        const bool pv = p < DNC * per;
        if (!pv) p = 0;
        const int cl = p / per, j = p - cl * per;
        const int tin = 2 * j + (odd ? 1 : 0);
        // g index (with +1 offset for the leading zero): odd: t = j (kt=1); even: t = j (kt=0), t = j-1 (kt=2)
        const float* gp = g + cl * 32 * GS + half * GS + j + 1;
        const float* wl = wd + half * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (odd) {
            float wb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wb[u] = wl[(1 * 32 + 2 * u) * 32];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[u], gp[2 * u * GS], acc, 0, 0, 0);
        } else {
            float wa[16], wc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { wa[u] = wl[(0 * 32 + 2 * u) * 32]; wc[u] = wl[(2 * 32 + 2 * u) * 32]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u], gp[2 * u * GS], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[u], gp[2 * u * GS - 1], acc, 0, 0, 0);
            }
        }
        if (pv && cl < nclips) {
            const size_t go = (size_t)(clip0 + cl) * 32 * TIN + tin;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = yin[go + c * TIN] > 0.f ? acc[r] : 0.f;
                gout[go + c * TIN] = v;
                if constexpr (TO_LDS) gx_lds[cl * 32 * GXS + c * GXS + tin + 1] = v;
            }
        }
    }
}

__global__ void __launch_bounds__(DNT)
snd_dgrad_kernel(int clip_lo, int clip_hi, const float* __restrict__ wd1, const float* __restrict__ wd2,
                 const float* __restrict__ wd3, const float* __restrict__ a1, const float* __restrict__ a2,
                 const float* __restrict__ a3, const float* __restrict__ g4, float* __restrict__ g3,
                 float* __restrict__ g2, float* __restrict__ g1) {
    __shared__ float lds[D_END];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip0 = clip_lo + blockIdx.x * DNC;
    const int nclips = (clip_hi - clip0) < DNC ? (clip_hi - clip0) : DNC;
    zero_lds<DNT>(lds, D_END, tid);
    __syncthreads();
    for (int e = tid; e < DNC * 32 * T4; e += DNT) {
        const int cl = e / (32 * T4), rem = e - cl * 32 * T4, n = rem / T4, t = rem - n * T4;
        if (cl < nclips) lds[D_G4 + cl * 32 * G4S + n * G4S + t + 1] = g4[(size_t)(clip0 + cl) * 32 * T4 + rem];
    }
    __syncthreads();
    snd_dgrad_layer<T3, T4, G4S, G3S, true>(lds + D_G4, lds + D_G3, wd3, a3, g3, clip0, nclips, wave, lane);
    __syncthreads();
    snd_dgrad_layer<T2, T3, G3S, G2S, true>(lds + D_G3, lds + D_G2, wd2, a2, g2, clip0, nclips, wave, lane);
    __syncthreads();
    snd_dgrad_layer<T1, T2, G2S, 1, false>(lds + D_G2, nullptr, wd1, a1, g1, clip0, nclips, wave, lane);
}

// ------------------------------------------------------------------------------------------
// wgrad: 9 waves, fixed roles; accumulators live in registers across the workgroup's clips
//   waves 0..6 : dW0[n][kb*32 .. +32)   A = g1[n][t],  B = x[(2t+kt)*41 + f]   (lane = k)
//   wave  7    : dW1[n][c][kt]          A = g2[n][t],  B = y1[c][2t+kt]        (lane = c)
//   wave  8    : dW2, dW3 likewise
// ------------------------------------------------------------------------------------------
constexpr int WNW = 9, WNT = WNW * 64;
// LDS (floats): x (padded rows), y1,y2,y3 [c][odd stride], g1..g4 [n][odd stride, zero-padded to even length]
constexpr int WY1S = 49, WY2S = 23, WY3S = 11;
constexpr int WG1S = 49, WG2S = 25, WG3S = 13, WG4S = 7;          // >= TOUT+1 (zero tail), odd
constexpr int W_X = 0, W_Y1 = W_X + XCLIP, W_Y2 = W_Y1 + 32 * WY1S, W_Y3 = W_Y2 + 32 * WY2S,
              W_G1 = W_Y3 + 32 * WY3S, W_G2 = W_G1 + 32 * WG1S, W_G3 = W_G2 + 32 * WG2S, W_G4 = W_G3 + 32 * WG3S,
              W_END = W_G4 + 32 * WG4S;

// two-phase row staging: all global loads first (registers), LDS stores later, so that the loads of
// the NEXT clip are in flight while the current clip is being multiplied
template <int TT, int NT>
struct RowRegs {
    static constexpr int N = (32 * TT + NT - 1) / NT;
    float v[N];
    __device__ __forceinline__ void load(const float* __restrict__ src, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int e = tid + i * NT;
            v[i] = e < 32 * TT ? src[e] : 0.f;
        }
    }
    template <int S>
    __device__ __forceinline__ void store(float* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int e = tid + i * NT;
            if (e < 32 * TT) dst[(e / TT) * S + (e % TT)] = v[i];
        }
    }
};

template <int NT>
struct ClipRegs {
    static constexpr int N = (T0 * F / 4 + NT - 1) / NT;
    float4 v[N];
    __device__ __forceinline__ void load(const float* __restrict__ src, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int e = tid + i * NT;
            v[i] = e < T0 * F / 4 ? ((const float4*)src)[e] : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(float* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int e = tid + i * NT;
            if (e < T0 * F / 4) {
                const int r = e / 10, f4 = e - r * 10;
                float* d = dst + r * XROW + f4 * 4;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        }
    }
};

template <int TOUT, int GS, int YS>
__device__ __forceinline__ void wgrad_taps(f32x16 (&acc)[3], float& bsum, const float* __restrict__ g,
                                           const float* __restrict__ yin, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    const float* ap = g + l31 * GS + half;
    const float* bp = yin + l31 * YS + 2 * half;
#pragma unroll
    for (int s = 0; s < (TOUT + 1) / 2; ++s) {
        const float a = ap[2 * s];
        bsum += a;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[4 * s + kt], acc[kt], 0, 0, 0);
    }
}

__global__ void __launch_bounds__(WNT)
snd_wgrad_kernel(int clip_lo, int clip_hi, int B, const float* __restrict__ pos, const float* __restrict__ neg,
                 const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ a3,
                 const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ g3,
                 const float* __restrict__ g4, float* __restrict__ slabs) {
    __shared__ float lds[W_END];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[3], acc2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; acc2[i][r] = 0.f; }
    float bsum = 0.f, bsum2 = 0.f;
    // per-lane B offset for the layer-0 tiles: k = wave*32 + lane -> (kt, f)
    int k0 = wave * 32 + l31;
    if (k0 >= 200) k0 = 0;
    const int xoff = (k0 / 40) * XROW + (k0 % 40) + 2 * half * XROW;
    zero_lds<WNT>(lds, W_END, tid);
    ClipRegs<WNT> rx;
    RowRegs<T1, WNT> ry1, rg1;
    RowRegs<T2, WNT> ry2, rg2;
    RowRegs<T3, WNT> ry3, rg3;
    RowRegs<T4, WNT> rg4;
    auto load_clip = [&](int clip) {
        const float* src = clip < B ? pos + (size_t)clip * T0 * F : neg + (size_t)(clip - B) * T0 * F;
        rx.load(src, tid);
        ry1.load(a1 + (size_t)clip * 32 * T1, tid);
        ry2.load(a2 + (size_t)clip * 32 * T2, tid);
        ry3.load(a3 + (size_t)clip * 32 * T3, tid);
        rg1.load(g1 + (size_t)clip * 32 * T1, tid);
        rg2.load(g2 + (size_t)clip * 32 * T2, tid);
        rg3.load(g3 + (size_t)clip * 32 * T3, tid);
        rg4.load(g4 + (size_t)clip * 32 * T4, tid);
    };
    if (clip_lo + (int)blockIdx.x < clip_hi) load_clip(clip_lo + blockIdx.x);
#pragma unroll 1
    for (int clip = clip_lo + blockIdx.x; clip < clip_hi; clip += gridDim.x) {
        __syncthreads();                          // previous clip's MFMAs are done with the LDS images
        rx.store(lds + W_X, tid);
        ry1.store<WY1S>(lds + W_Y1, tid);
        ry2.store<WY2S>(lds + W_Y2, tid);
        ry3.store<WY3S>(lds + W_Y3, tid);
        rg1.store<WG1S>(lds + W_G1, tid);
        rg2.store<WG2S>(lds + W_G2, tid);
        rg3.store<WG3S>(lds + W_G3, tid);
        rg4.store<WG4S>(lds + W_G4, tid);
        __syncthreads();
        if (clip + (int)gridDim.x < clip_hi) load_clip(clip + gridDim.x);   // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        if (wave < 7) {
            const float* ap = lds + W_G1 + l31 * WG1S + half;
            const float* bp = lds + W_X + xoff;
#pragma unroll 8
            for (int s = 0; s < T1 / 2; ++s) {
                const float a = ap[2 * s];
                bsum += a;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[4 * s * XROW], acc[0], 0, 0, 0);
            }
        } else if (wave == 7) {
            wgrad_taps<T2, WG2S, WY1S>(acc, bsum, lds + W_G2, lds + W_Y1, lane);
        } else {
            wgrad_taps<T3, WG3S, WY2S>(acc, bsum, lds + W_G3, lds + W_Y2, lane);
            wgrad_taps<T4, WG4S, WY3S>(acc2, bsum2, lds + W_G4, lds + W_Y3, lane);
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * SND_SLICE;
    bsum += __shfl_down(bsum, 32, 64);
    bsum2 += __shfl_down(bsum2, 32, 64);
    if (wave < 7) {
        const int k = wave * 32 + l31;
        if (k < 200) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[n * 200 + k] = acc[0][r];
            }
        }
        if (wave == 0 && half == 0) slab[6400 + l31] = bsum;
    } else {
        float* sl = slab + 6432 + (wave == 7 ? 0 : 3104);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                sl[(n * 32 + l31) * 3 + kt] = acc[kt][r];
                if (wave == 8) sl[3104 + (n * 32 + l31) * 3 + kt] = acc2[kt][r];
            }
        if (half == 0) {
            sl[3072 + l31] = bsum;
            if (wave == 8) sl[3104 + 3072 + l31] = bsum2;
        }
    }
}

// 32 consecutive elements x 8 slices of the slab index per block: a lane sums every 8th slab with all its loads
// in flight at once (one memory latency instead of G/8), the 8 partial sums are folded in a fixed order.
__global__ void __launch_bounds__(256)
snd_reduce_kernel(const float* __restrict__ slabs, int G, float* __restrict__ out) {
    __shared__ float part[8][33];
    const int l32 = threadIdx.x & 31, gs = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + l32;
    const bool live = e < SND_SLICE;
    float s = 0.f;
    if (live) {
#pragma unroll 16
        for (int g = gs; g < G; g += 8) s += slabs[(size_t)g * SND_SLICE + e];
    }
    part[gs][l32] = s;
    __syncthreads();
    if (gs == 0 && live)
        out[e] = ((part[0][l32] + part[1][l32]) + (part[2][l32] + part[3][l32])) +
                 ((part[4][l32] + part[5][l32]) + (part[6][l32] + part[7][l32]));
}
}  // namespace

static const int kSndG = 128;
size_t snd_slab_floats() { return (size_t)kSndG * SND_SLICE; }

int launch_snd_fwd(var_ctx* c, hipStream_t s, const float* params, const float* pos, const float* neg, int B) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    const int lo = pos ? 0 : B, hi = neg ? 2 * B : B;
    if (hi <= lo) return VAR_OK;
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             F_END * 4));
        attr_set |= var_dev_bit(c);
    }
    ProfScope prof(c, s, TAG_SND_FWD);
    hipLaunchKernelGGL(snd_fwd_kernel, dim3((hi - lo + FNC - 1) / FNC), dim3(FNT), F_END * 4, s, pos, neg, B, lo, hi,
                       c->wpack + K.snd_f[0], c->wpack + K.snd_f[1], c->wpack + K.snd_f[2], c->wpack + K.snd_f[3],
                       params + L.snd_b[0], params + L.snd_b[1], params + L.snd_b[2], params + L.snd_b[3],
                       c->sact[1], c->sact[2], c->sact[3], c->sact[4]);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// clips [lo,hi) of the (pos | neg) stack took part in the forward
int launch_snd_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    const int lo = c->saved_pos ? 0 : B, hi = c->saved_neg ? 2 * B : B;
    if (hi <= lo) {
        { const int rz = var_zero_async(c, s, grads + L.snd_w[0], sizeof(float) * SND_SLICE); if (rz != VAR_OK) return rz; }
        return VAR_OK;
    }
    {
        ProfScope prof(c, s, TAG_SND_DGRAD);
        hipLaunchKernelGGL(snd_dgrad_kernel, dim3((hi - lo + DNC - 1) / DNC), dim3(DNT), 0, s, lo, hi,
                           c->wpack + K.snd_d[1], c->wpack + K.snd_d[2], c->wpack + K.snd_d[3],
                           c->sact[1], c->sact[2], c->sact[3], c->gsact[4], c->gsact[3], c->gsact[2], c->gsact[1]);
    }
    const int G = hi - lo < kSndG ? hi - lo : kSndG;
    float* slabs = c->slabs + c->snd_slab_off;
    {
        ProfScope prof(c, s, TAG_SND_WGRAD);
        hipLaunchKernelGGL(snd_wgrad_kernel, dim3(G), dim3(WNT), 0, s, lo, hi, B, c->saved_pos, c->saved_neg,
                           c->sact[1], c->sact[2], c->sact[3], c->gsact[1], c->gsact[2], c->gsact[3], c->gsact[4], slabs);
    }
    ProfScope prof(c, s, TAG_SND_REDUCE);
    hipLaunchKernelGGL(snd_reduce_kernel, dim3((SND_SLICE + 31) / 32), dim3(256), 0, s, slabs, G,
                       grads + L.snd_w[0]);
    VAR_HIP_CHECK(c, hipGetLastError());
    (void)params;
    return VAR_OK;
}
