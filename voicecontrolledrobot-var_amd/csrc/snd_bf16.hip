// The 11x5 stride-2 convolution of the iTHOR sound CNN (models/pretext/ai2thor_pretext_model.py:24-26, conv 2:
// 64 -> 64 channels, (n,64,300,20) -> (n,64,150,13), 45 % of the model's arithmetic) on v_mfma_f32_32x32x16_bf16 --
// the bf16 mode's own kernels (BASELINE config 4), replacing the gather-GEMM instances of gg.h for this layer.
//
// The gather-GEMM loads every operand element once per USE (an input element of this layer is used by 55/4 taps): in
// bf16 its matrix instructions are 16x shorter and the kernel ends up bound by the load path (6 wave-loads per MFMA).
// Here the input patch of a tile is staged ONCE into LDS, as bf16, and every tap reads it from there:
//
//   HBM   activations in "C8" form: (n, C/8, H, W, 8) bf16 -- a pixel's 8 channels are the 16 bytes one lane of the
//         MFMA supplies (k = 8h .. 8h+7), a plane row is one contiguous run.
//   tile  one clip x 38|37 output rows x 13 columns = <= 512 pixel slots = 16 MFMA column blocks, 4 per wave; both
//         32-row blocks of the 64 output channels: 8 accumulators (128 registers) per wave.  4 tiles per clip, one
//         workgroup (4 waves, one per SIMD) per CU, persistent.
//   K     4 quarters of 16 input channels x 55 taps; one quarter of the patch (85 rows) is in LDS while the next one
//         is being loaded (register-staged, double-buffered: 2 x 70.7 KB).
//   LDS   per quarter [k half h][column parity][row][13 slots of 16 B]: a stride-2 tap walks consecutive 16-byte
//         slots when the lanes walk the output row (ds_read_b128, conflict-free), the three/two zero slots of a
//         sub-row are the padding columns of this row AND of the next one, so a tap is base + IMMEDIATE offset: no
//         bounds test, no address arithmetic in the loop.
//   W     re-packed per step into fragment order (quarter, tap, channel block, lane): one 16-byte load per lane and
//         MFMA row block, straight from L2 into registers, one filter row ahead.
#include "var_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

namespace {

constexpr int CI = 64, CO = 64, HI = 300, WI = 20, HO = 150, WO = 13, KH = 11, KW = 5, NTAP = KH * KW;
constexpr int NQ = 4;                         // quarters of 16 input channels
constexpr int SLOT = 16;                      // bytes: 8 bf16 channels of one pixel
constexpr int SUBP = 13 * SLOT;               // 208: sub-row pitch (10 data slots + 3 zero slots shared with the next row)
constexpr int NR = 85;                        // patch rows of a 38-row tile
constexpr int PARB = (NR + 6) * SUBP;         // one column-parity image (+ rows that only unused pixel slots touch)
constexpr int PLANEB = 2 * PARB;              // one k-half
constexpr int BUFB = 2 * PLANEB;              // one quarter
constexpr int LDSB = 2 * BUFB + 8 * SUBP;
static_assert(LDSB <= 160 * 1024, "LDS");
constexpr int TILES = 4;
__device__ __constant__ int kRow0[TILES] = {0, 38, 76, 113};
__device__ __constant__ int kRows[TILES] = {38, 38, 37, 37};

__device__ __forceinline__ unsigned bf16_bits(float x) {          // round to nearest even (no NaNs in this model)
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// fp32 NCHW (n, C, HW) -> bf16 C8 (n, C/8, HW, 8)
__global__ void __launch_bounds__(256) to_c8_kernel(const float* __restrict__ x, uint4* __restrict__ y, long total, int C8, int HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pix = (int)(i % HW);
    const long pl = i / HW;                                       // n * C8 + plane
    const float* src = x + (pl * 8) * HW + pix;
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf16_bits(src[(long)j * HW]);
    y[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// OIHW fp32 (64,64,11,5) -> fragment order [q][tap][cb][lane][8]: lane (r, h) of block cb holds
// W[co = 32 cb + r][ci = 16 q + 8 h + j][tap], j = 0..7
__global__ void __launch_bounds__(256) pack_w2_kernel(const float* __restrict__ w, uint4* __restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NQ * NTAP * 2 * 64) return;
    const int lane = i & 63, cb = (i >> 6) & 1, qt = i >> 7, tap = qt % NTAP, q = qt / NTAP;
    const int co = 32 * cb + (lane & 31), ci0 = 16 * q + 8 * (lane >> 5);
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf16_bits(w[((long)co * CI + ci0 + j) * NTAP + tap]);
    wp[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// ---- forward --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) snd2_fwd_kernel(const uint4* __restrict__ x8, const uint4* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ y, int nclips) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p31 = lane & 31;
    const int ntiles = nclips * TILES;

    for (int i = tid; i < LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // this lane's four pixel slots: byte offset of (input row 2 oyl, slot ox) in its k-half's images
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int P = 128 * wave + 32 * m + p31, oyl = P / WO, ox = P - oyl * WO;
        abase[m] = h * PLANEB + oyl * 2 * SUBP + ox * SLOT;
    }

    // staging: a quarter of a tile's patch = 2 planes x nrows x 20 slots, 14 slots per thread through registers.
    // Column c of the map sits at parity (c + 5) & 1, slot (c + 5) >> 1 of its sub-row.
    constexpr int NST = 14;
    uint4 sreg[NST];
    auto stage_load = [&](int tile, int q) {
        const int clip = tile >> 2, t = tile & 3;
        const int y0 = 2 * kRow0[t] - 5, nrows = 2 * kRows[t] + 9;
        const uint4* src = x8 + ((long)clip * 8 + 2 * q) * (HI * WI);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + 256 * k;
            const int hh = e >= nrows * WI ? 1 : 0, e2 = e - hh * nrows * WI;
            const int i = e2 / WI, yy = y0 + i;
            const bool ok = e2 < nrows * WI && (unsigned)yy < (unsigned)HI;
            sreg[k] = ok ? src[(long)hh * (HI * WI) + (long)y0 * WI + e2] : make_uint4(0, 0, 0, 0);
        }
    };
    auto stage_store = [&](int tile, int buf) {
        const int t = tile & 3;
        const int nrows = 2 * kRows[t] + 9;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + 256 * k;
            const int hh = e >= nrows * WI ? 1 : 0, e2 = e - hh * nrows * WI;
            const int i = e2 / WI, c5 = e2 - i * WI + 5;
            if (e2 < nrows * WI)
                *(uint4*)(lds + buf * BUFB + hh * PLANEB + (c5 & 1) * PARB + i * SUBP + (c5 >> 1) * SLOT) = sreg[k];
        }
    };
    static_assert(2 * NR * WI <= NST * 256, "staging registers");

    int tile = blockIdx.x;
    if (tile < ntiles) { stage_load(tile, 0); stage_store(tile, 0); }
    __syncthreads();

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16_t acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][cb][r] = 0.f;

#pragma unroll 1
        for (int q = 0; q < NQ; ++q) {
            const int buf = q & 1;
            const int ntile = q < NQ - 1 ? tile : tile + (int)gridDim.x, nq = (q + 1) & 3;
            const bool have = ntile < ntiles;
            if (have) stage_load(ntile, nq);
            __builtin_amdgcn_sched_barrier(0);

            const unsigned char* img = lds + buf * BUFB;
            const uint4* wq = wp + (long)q * NTAP * 128 + lane;
            // software pipeline, pinned with sched_barriers (left alone, hipcc sinks every load to its use and waits
            // for it there): filter fragments one filter ROW ahead, pixel fragments one TAP ahead
            uint4 wrow[2][KW][2];
            bf16x8_t a[2][4];
            auto toff = [](int tap) { const int ky = tap / KW, kx = tap - ky * KW; return ky * SUBP + (kx & 1) * PARB + (kx >> 1) * SLOT; };
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) { wrow[0][kx][0] = wq[kx * 128]; wrow[0][kx][1] = wq[kx * 128 + 64]; }
#pragma unroll
            for (int m = 0; m < 4; ++m) a[0][m] = *(const bf16x8_t*)(img + abase[m] + toff(0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const int cur = ky & 1;
                if (ky + 1 < KH) {
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) {
                        wrow[cur ^ 1][kx][0] = wq[((ky + 1) * KW + kx) * 128];
                        wrow[cur ^ 1][kx][1] = wq[((ky + 1) * KW + kx) * 128 + 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const int tap = ky * KW + kx, ac = tap & 1;
                    if (tap + 1 < NTAP) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) a[ac ^ 1][m] = *(const bf16x8_t*)(img + abase[m] + toff(tap + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8_t w0 = __builtin_bit_cast(bf16x8_t, wrow[cur][kx][0]);
                    const bf16x8_t w1 = __builtin_bit_cast(bf16x8_t, wrow[cur][kx][1]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a[ac][m], acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a[ac][m], acc[m][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (have) stage_store(ntile, buf ^ 1);
            __syncthreads();
        }

        // bias + ReLU, fp32 NCHW: lanes walk the pixels (128 contiguous bytes per channel and block)
        const int clip = tile >> 2, t = tile & 3;
        const int npx = kRows[t] * WO;
        float* yo = y + (long)clip * CO * (HO * WO) + kRow0[t] * WO;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int P = 128 * wave + 32 * m + p31;
            if (P < npx) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * h;
                        yo[(long)co * (HO * WO) + P] = fmaxf(acc[m][cb][r] + bias[co], 0.f);
                    }
            }
        }
    }
}

}  // namespace

// workspace (bytes) of the bf16 kernels for up to `nclips` clips: [x8 | wp]
long snd_bf16_workspace_bytes(int nclips) {
    return (long)nclips * CI * HI * WI * 2 + (long)NQ * NTAP * 2 * 64 * 16 + 256;
}

int snd2_bf16_fwd(var_ctx* c, hipStream_t s, const float* x, const float* w, const float* bias, float* y, int nclips, void* ws) {
    uint4* x8 = (uint4*)ws;
    uint4* wp = (uint4*)((char*)ws + (((long)nclips * CI * HI * WI * 2 + 255) & ~255L));
    const long total = (long)nclips * 8 * HI * WI;
    hipLaunchKernelGGL(to_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x8, total, 8, HI * WI);
    VAR_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(pack_w2_kernel, dim3((NQ * NTAP * 128 + 255) / 256), dim3(256), 0, s, w, wp);
    VAR_HIP_CHECK(c, hipGetLastError());
    static bool attr = false;
    if (!attr) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd2_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
        attr = true;
    }
    const int ntiles = nclips * TILES;
    hipLaunchKernelGGL(snd2_fwd_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(256), LDSB, s, x8, wp, bias, y, nclips);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
