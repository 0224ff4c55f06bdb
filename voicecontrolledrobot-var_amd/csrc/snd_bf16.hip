// The sound CNN of the iTHOR model (models/pretext/ai2thor_pretext_model.py:22-28: conv 11x11 s2 1 -> 64, conv 11x5 s2
// 64 -> 64 -- 45 % of the model's arithmetic --, conv 7x3 s2 64 -> 64) in the model's bf16 mode (BASELINE config 4), all three
// layers in all three directions on v_mfma_f32_32x32x16_bf16, replacing the gather-GEMM instances of gg.h.
//
// The gather-GEMM loads every operand element once per USE (an input element of conv 2 is used by 55/4 taps): in bf16 its
// matrix instructions are 16x shorter and the kernel ends up bound by the load path (6 wave-loads per MFMA).  Here the
// input patch of a tile is staged ONCE into LDS, as bf16, and every tap reads it from there.  The activations between
// the layers live as "C8" bf16 images: (clip, C/8, H, W, 8) -- a pixel's 8 channels are the 16 bytes one lane of the MFMA
// supplies (k = 8h .. 8h+7), a plane row is one contiguous run -- each written by the kernel that produces the map
// (conv 1's and conv 2's forward stores; conv 3's data-gradient store for the gradient image conv 2's backward reads),
// together with one 32-bit word of ReLU signs per (pixel, lane half) in the order the data-gradient stores want them.
//
// Forward (conv 2 and 3: snd_fwd_kernel<Geo>, the geometry is a type):
//   tile  one clip x 38|37 output rows x 13 columns (conv 3: the whole 73 x 7 map) = <= 512 pixel slots = 16 MFMA column
//         blocks, 4 per wave; both 32-row blocks of the 64 output channels: 8 accumulators (128 registers) per wave.  One
//         workgroup (4 waves, one per SIMD) per CU, persistent.
//   K     4 quarters of 16 input channels x all taps; one quarter of the patch is in LDS while the next one is being
//         loaded (register-staged, double-buffered: 2 x 70 KB).
//   LDS   per quarter [k half h][column parity][row][slots of 16 B]: a stride-2 tap walks consecutive 16-byte slots when
//         the lanes walk the output row (ds_read_b128), the zero slots of a sub-row are the padding columns of this row
//         AND of the next one, so a tap is base + IMMEDIATE offset: no bounds test, no address arithmetic in the loop.
//   W     re-packed per step into fragment order (quarter, tap, channel block, lane): one 16-byte load per lane and
//         MFMA row block, straight from L2 into registers, one filter row ahead.
// Data gradient (snd_dgrad_kernel<DGeo>), weight gradient (snd_wgrad_kernel<WGeo>) and conv 1 (snd1_*): see their sections.
#include "var_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// 16 bytes per lane at (wave-uniform byte offset) + (lane offset) of a buffer: scalar base + one offset register instead of
// a 64-bit address pair per load (hipcc otherwise keeps one pair per 4 KB of a table alive across the whole kernel)
__device__ __forceinline__ u32x4_t wload(__amdgpu_buffer_rsrc_t r, int lane_off, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, byte_off, 0);
}

PH_DECL();
#ifdef VAR_PHASES
extern "C" int var_debug_phases_snd(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

namespace {

constexpr int CI = 64, CO = 64, HI = 300, WI = 20, HO = 150, WO = 13;      // conv 2 (the conversion kernels' callers)
constexpr int NQ = 4;                         // quarters of 16 input channels
constexpr int SLOT = 16;                      // bytes: 8 bf16 channels of one pixel

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two floats -> two bf16 in one register (v_cvt_pk_bf16_f32: round to nearest even, as bf16_bits below)
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
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

// the same for a 64-channel map, one thread per pixel, plus the sign masks the data-gradient kernel's store wants:
// mask[(n*HW + pix)*2 + h] bit 16 cb + 4 g + j = x[n][32 cb + 8 g + 4 h + j][pix] > 0 (as bf16)
__global__ void __launch_bounds__(256) to_c8_mask_kernel(const float* __restrict__ x, uint4* __restrict__ y,
                                                         unsigned* __restrict__ mask, long total, int HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pix = (int)(i % HW);
    const long n = i / HW;
    const float* src = x + n * 64 * HW + pix;
    unsigned mk[2] = {0u, 0u};
#pragma unroll
    for (int pl = 0; pl < 8; ++pl) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = bf16_bits(src[(long)(8 * pl + j) * HW]);
            if (v[j] - 1u < 0x7fffu) mk[j >> 2] |= 1u << (16 * (pl >> 2) + 4 * (pl & 3) + (j & 3));
        }
        y[(n * 8 + pl) * HW + pix] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
    mask[2 * i] = mk[0];
    mask[2 * i + 1] = mk[1];
}

// ---- forward --------------------------------------------------------------------------------------------------
// One kernel for the model's two 64 -> 64 stride-2 sound convolutions: the geometry is a type.
struct Geo2 {            // conv 2: 11x5, pad 5, (300,20) -> (150,13); 4 tiles of 38|38|37|37 output rows per clip
    static constexpr int HI = 300, WI = 20, HO = 150, WO = 13, KH = 11, KW = 5, PH = 5, PW = 5, NSLOT = 13, TILES = 4, TROWS = 38;
    static constexpr bool SEQ = false;
    __device__ static int row0(int t) { return t < 2 ? 38 * t : 76 + 37 * (t - 2); }
    __device__ static int rows(int t) { return t < 2 ? 38 : 37; }
};
struct Geo3 {            // conv 3: 7x3, pad 1, (150,13) -> (73,7) = 511 pixels: one tile per clip; stored as the GRU's sequence
    static constexpr int HI = 150, WI = 13, HO = 73, WO = 7, KH = 7, KW = 3, PH = 1, PW = 1, NSLOT = 7, TILES = 1, TROWS = 73;
    static constexpr bool SEQ = true;
    __device__ static int row0(int) { return 0; }
    __device__ static int rows(int) { return 73; }
};
template <class G>
struct FwdLayout {
    static constexpr int NTAP = G::KH * G::KW;
    static constexpr int SUBP = G::NSLOT * SLOT;                  // sub-row pitch: data slots + zero slots shared with the next row
    static constexpr int NR = 2 * G::TROWS + G::KH - 2;           // patch rows of the largest tile
    static constexpr int PARB = (NR + 6) * SUBP;                  // one column-parity image (+ rows only unused pixel slots touch)
    static constexpr int PLANEB = 2 * PARB, BUFB = 2 * PLANEB, LDSB = 2 * BUFB + 8 * SUBP;
    static constexpr int NST = (2 * NR * G::WI + 255) / 256;      // staging slots per thread and quarter
    static_assert(LDSB <= 160 * 1024, "LDS");
    static_assert(G::TROWS * G::WO <= 512, "16 MFMA column blocks per tile");
    // the zero slots: column c sits at parity (c + PW) & 1, slot (c + PW) >> 1; a tap reads slot ox + (kx >> 1) <= NSLOT + 1
    static_assert((G::WO - 1) + ((G::KW - 1) >> 1) <= G::NSLOT + 1, "padding slots");
};

// OIHW fp32 (64,64,KH,KW) -> fragment order [q][tap][cb][lane][8]: lane (r, h) of block cb holds
// W[co = 32 cb + r][ci = 16 q + 8 h + j][tap] (forward), or, T, W[co = 16 q + 8 h + j][ci = 32 cb + r][tap] (data gradient)
template <bool T>
__global__ void __launch_bounds__(256) pack_w_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int ntap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NQ * ntap * 2 * 64) return;
    const int lane = i & 63, cb = (i >> 6) & 1, qt = i >> 7, tap = qt % ntap, q = qt / ntap;
    const int r = 32 * cb + (lane & 31), k0 = 16 * q + 8 * (lane >> 5);
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf16_bits(T ? w[((long)(k0 + j) * CI + r) * ntap + tap] : w[((long)r * CI + k0 + j) * ntap + tap]);
    wp[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// y: bias + ReLU, fp32 NCHW or (SEQ) the (clip, oy, co*WO + ox) sequence; y8 / ymask (optional): the same values as the next
// layer's C8 bf16 image and the sign words its data-gradient kernel stores through (to_c8_mask_kernel's layouts)
// (F32: the fp32 copy of a non-sequence output exists; y8 / ymask exist exactly when the output is not the sequence --
//  compile-time, as run-time null checks they were a scalar branch in front of every store)
template <class G, bool F32>
__global__ void __launch_bounds__(256) snd_fwd_kernel(const uint4* __restrict__ x8, const uint4* __restrict__ wp,
                                                      const float* __restrict__ bias, float* __restrict__ y, uint2* __restrict__ y8,
                                                      unsigned* __restrict__ ymask, int nclips) {
    using L = FwdLayout<G>;
    constexpr int WO = G::WO, KH = G::KH, KW = G::KW, NTAP = L::NTAP, SUBP = L::SUBP, PARB = L::PARB, PLANEB = L::PLANEB, BUFB = L::BUFB;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p31 = lane & 31;
    const int ntiles = nclips * G::TILES;

    for (int i = tid; i < L::LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // this lane's four pixel slots: byte offset of (input row 2 oyl, slot ox) in its k-half's images
    int abase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int P = 128 * wave + 32 * m + p31, oyl = P / WO, ox = P - oyl * WO;
        abase[m] = h * PLANEB + oyl * 2 * SUBP + ox * SLOT;
    }

    // staging: a quarter of a tile's patch = 2 planes x NR rows x WI slots through registers (every tile stages the NR rows
    // of the largest one: a slot's plane / row / column are then per-thread constants, computed once -- with one wave per SIMD
    // the per-slot divisions of every quarter were paid in full).  Column c of the map sits at parity (c + PW) & 1, slot
    // (c + PW) >> 1 of its sub-row.
    constexpr int NST = L::NST;
    uint4 sreg[NST];
    int s_src[NST], s_dst[NST], s_row[NST];                     // source slot within the quarter image (from row y0), LDS offset, patch row
    unsigned sok = 0u;                                          // which of the slots in flight lie inside the map
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int e = tid + 256 * k, hh = e >= L::NR * G::WI ? 1 : 0, e2 = e - hh * L::NR * G::WI, i = e2 / G::WI, c5 = e2 - i * G::WI + G::PW;
        const bool in = e < 2 * L::NR * G::WI;
        s_src[k] = hh * (G::HI * G::WI) + e2;
        s_dst[k] = in ? hh * PLANEB + (c5 & 1) * PARB + i * SUBP + (c5 >> 1) * SLOT : -1;
        s_row[k] = in ? i : -100000;
    }
    auto stage_load = [&](int tile, int q) {
        const int clip = tile / G::TILES, t = tile - clip * G::TILES;
        const int y0 = 2 * G::row0(t) - G::PH;
        const uint4* src = x8 + ((long)clip * 8 + 2 * q) * (G::HI * G::WI) + (long)y0 * G::WI;
        // (loads from a clamped address; the zeroing waits for the LDS store: a select on the loaded value made every load wait)
        sok = 0u;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const bool ok = (unsigned)(y0 + s_row[k]) < (unsigned)G::HI;
            sreg[k] = src[ok ? s_src[k] : 5 * G::WI];
            sok |= ok ? 1u << k : 0u;
        }
    };
    auto stage_store = [&](int, int buf) {
#pragma unroll
        for (int k = 0; k < NST; ++k)
            if (s_dst[k] >= 0) *(uint4*)(lds + buf * BUFB + s_dst[k]) = (sok >> k) & 1u ? sreg[k] : make_uint4(0, 0, 0, 0);
    };

    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, NQ * NTAP * 2048, 0x00020000);
    int tile = blockIdx.x;
    if (tile < ntiles) { stage_load(tile, 0); stage_store(tile, 0); }
    __syncthreads();
    PH_INIT(0);

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
            PH(4);
            if (have) stage_load(ntile, nq);
            __builtin_amdgcn_sched_barrier(0);
            PH(0);

            const unsigned char* img = lds + buf * BUFB;
            const int wq0 = q * NTAP * 2048;
            // software pipeline, pinned with sched_barriers (left alone, hipcc sinks every load to its use and waits
            // for it there): filter fragments one filter ROW ahead, pixel fragments one TAP ahead.  (Two rows ahead, and the
            // next quarter's first rows loaded across the staging barrier, changed nothing: 465 -> 463 us.)
            u32x4_t wrow[2][KW][2];
            bf16x8_t a[2][4];
            auto toff = [](int tap) { const int ky = tap / KW, kx = tap - ky * KW; return ky * SUBP + (kx & 1) * PARB + (kx >> 1) * SLOT; };
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) { wrow[0][kx][0] = wload(wr, lane * 16, wq0 + kx * 2048); wrow[0][kx][1] = wload(wr, lane * 16, wq0 + kx * 2048 + 1024); }
#pragma unroll
            for (int m = 0; m < 4; ++m) a[0][m] = *(const bf16x8_t*)(img + abase[m] + toff(0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const int cur = ky & 1;
                if (ky + 1 < KH) {
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) {
                        wrow[cur ^ 1][kx][0] = wload(wr, lane * 16, wq0 + ((ky + 1) * KW + kx) * 2048);
                        wrow[cur ^ 1][kx][1] = wload(wr, lane * 16, wq0 + ((ky + 1) * KW + kx) * 2048 + 1024);
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
            PH(1);
            if (have) stage_store(ntile, buf ^ 1);
            PH(2);
            __syncthreads();
            PH(3);
        }

        // bias + ReLU; lanes walk the pixels (NCHW: 128 contiguous bytes per channel and block)
        const int clip = tile / G::TILES, t = tile - clip * G::TILES;
        const int npx = G::rows(t) * WO, pix0 = G::row0(t) * WO;
        float bv[2][16];                                        // this lane's 32 biases, in one batch (one by one in front of each
#pragma unroll                                                  // store they cost a quarter of the kernel)
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[cb][r] = bias[32 * cb + 8 * (r >> 2) + 4 * h + (r & 3)];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            int P = 128 * wave + 32 * m + p31;
            asm volatile("" : "+v"(P));                         // (tile-invariant: hipcc would keep every store offset alive across tiles)
            if (P < npx) {
                unsigned mk = 0u;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int co = 32 * cb + 8 * g + 4 * h + e;
                            v[e] = fmaxf(acc[m][cb][4 * g + e] + bv[cb][4 * g + e], 0.f);
                            if constexpr (G::SEQ) {
                                const int oyl = P / WO, ox = P - oyl * WO;
                                y[((long)clip * G::HO + oyl) * (CO * WO) + co * WO + ox] = v[e];
                            } else if constexpr (F32) {
                                y[((long)clip * CO + co) * (G::HO * WO) + pix0 + P] = v[e];
                            }
                        }
                        if constexpr (!G::SEQ) {       // plane 4 cb + g, this lane's half (4 h .. 4 h + 3) of the pixel's 16-byte slot
                            const unsigned p01 = pack_bf16(v[0], v[1]), p23 = pack_bf16(v[2], v[3]);
                            y8[(((long)clip * 8 + 4 * cb + g) * (G::HO * WO) + pix0 + P) * 2 + h] = make_uint2(p01, p23);
                            mk |= ((p01 & 0xffffu ? 1u : 0u) | (p01 >> 16 ? 2u : 0u) | (p23 & 0xffffu ? 4u : 0u) | (p23 >> 16 ? 8u : 0u))
                                  << (16 * cb + 4 * g);
                        }
                    }
                if constexpr (!G::SEQ) ymask[((long)clip * (G::HO * WO) + pix0 + P) * 2 + h] = mk;
            }
        }
    }
}

// ---- data gradient ---------------------------------------------------------------------------------------------
// dx[ci][iy][ix] = sum_{co,ky,kx} gy[co][(iy+PH-ky)/2][(ix+PW-kx)/2] W[co][ci][ky][kx] over the taps of matching parity:
// the four pixel-parity classes (a, b) = (iy & 1, ix & 1) are four stride-1 correlations of the gy map, dy = (a+PH-ky)/2,
// dx = (b+PW-kx)/2 -- rows outside the map are staged as zeros, the column never leaves the map (+ one zero slot).
// Tile: one clip x ROWS rows u x V column pairs v; wave (wm, cb) owns MB column blocks and one 32-channel row block for
// BOTH column parities b: a pixel pair (2v, 2v+1) ends up in one lane, so a map row is stored as 8 contiguous bytes per
// lane (4-byte stores at an 8-byte stride, one parity at a time, made the first version 4x slower than its arithmetic).
// The two row parities a are two passes over the same gy patch (all 64 channels: 8 planes of 16-byte slots,
// double-buffered).  A filter row's taps read (KW+1)/2 distinct pixel fragments.  The ReLU mask of the layer below
// comes as one 32-bit word per (pixel, lane half) in this epilogue's register order (written by the forward / the
// conversion kernel); the store also yields the channel sums of dx (the bias gradient of the layer below) and, on
// request, dx as the C8 bf16 image the layer below's own gradient kernels read.
struct DGeo2 {           // conv 2: dx (300,20) from gy (150,13); 6 tiles of 25 rows u
    static constexpr int HI = 300, WI = 20, HO = 150, WO = 13, KH = 11, KW = 5, PH = 5, PW = 5;
    static constexpr int ROWS = 25, TILES = 6, MB = 4, NSLOT = 13;
    static constexpr bool OUT8 = false, OUT16 = true;      // besides the optional fp32 dx: bf16 NCHW for conv 1's weight gradient
};
struct DGeo3 {           // conv 3: dx (150,13) from gy (73,7); 3 tiles of 25 rows u; the 7th pixel pair of a row is half empty
    static constexpr int HI = 150, WI = 13, HO = 73, WO = 7, KH = 7, KW = 3, PH = 1, PW = 1;
    static constexpr int ROWS = 25, TILES = 3, MB = 3, NSLOT = 8;
    static constexpr bool OUT8 = true, OUT16 = false;      // the C8 image = conv 2's gy
};
template <class G>
struct DgLayout {
    static constexpr int NTAP = G::KH * G::KW, V = (G::WI + 1) / 2, NSET = (G::KW + 1) / 2, DXMAX = (1 + G::PW) / 2;
    static constexpr int DYMIN = (G::PH - (G::KH - 2)) / 2;       // a = 0, the last odd ky (exact: PH and KH are odd): -2 for both layers
    static constexpr int DYMAX = (1 + G::PH) / 2;
    static constexpr int PR = G::ROWS + DYMAX - DYMIN, SUBP = G::NSLOT * SLOT, PLB = (PR + 3) * SUBP, BUFB = 8 * PLB, LDSB = 2 * BUFB;
    static constexpr int NSL = 8 * PR * G::WO, NST = (NSL + 255) / 256;
    static_assert(G::PH % 2 == 1 && G::PW % 2 == 1 && DYMIN == -2, "tap parities");
    static_assert(2 * 32 * G::MB >= G::ROWS * V && G::ROWS * G::TILES * 2 >= G::HI, "tile");
    static_assert((V - 1) + DXMAX < G::NSLOT + (G::NSLOT > G::WO ? 0 : 1) || G::NSLOT > G::WO, "zero slot");
};

// (which outputs exist is compile-time: as run-time null checks they were two scalar branches in front of every store)
template <class G, int A, bool F32>
__device__ __forceinline__ void dgrad_pass(const unsigned char* __restrict__ img, const int (&abase)[G::MB], __amdgpu_buffer_rsrc_t wq,
                                           int wlane, const unsigned* __restrict__ mask, int cb, float* __restrict__ dxo,
                                           uint2* __restrict__ dx8, unsigned short* __restrict__ dx16, int u0, int wm, int p31, int h,
                                           float (&bsum)[16]) {
    using L = DgLayout<G>;
    constexpr int KW = G::KW, MB = G::MB, NSET = L::NSET, NKY = A ? (G::KH + 1) / 2 : G::KH / 2, NROW = NQ * NKY;
    f32x16_t acc[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][b][r] = 0.f;
    // the pass's mask words first: their latency runs under the matrix loop (loaded in the epilogue, one pixel block at a
    // time, they and the stores behind them took as long as the loop itself)
    constexpr bool EVEN = G::WI % 2 == 0;
    unsigned mw[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int P = 32 * MB * wm + 32 * m + p31, ul = P / L::V, v = P - ul * L::V, iy = 2 * (u0 + ul) + A;
        const bool live = P < G::ROWS * L::V && iy < G::HI, two = EVEN || 2 * v + 1 < G::WI;
        const int pix = live ? iy * G::WI + 2 * v : 0;
        mw[m][0] = mask[2 * pix + h];
        mw[m][1] = mask[2 * pix + (two ? 2 : 0) + h];
    }
    // row R = (q, i): ky = 2 i + 1 - A
    auto wofs = [](int R, int kx) { const int q = R / NKY, i = R - q * NKY; return (q * L::NTAP + (2 * i + 1 - A) * KW + kx) * 2048; };
    auto aofs = [](int R, int d) {              // pixel fragment set d: dx = DXMAX - d
        const int q = R / NKY, i = R - q * NKY, ky = 2 * i + 1 - A;
        return 2 * q * L::PLB + ((A + G::PH - ky) / 2 - L::DYMIN) * L::SUBP + (L::DXMAX - d) * SLOT;
    };
    u32x4_t wrow[3][KW];
    bf16x8_t a[2][NSET][MB];
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) wrow[R][kx] = wload(wq, wlane, wofs(R, kx));
#pragma unroll
    for (int d = 0; d < NSET; ++d)
#pragma unroll
        for (int m = 0; m < MB; ++m) a[0][d][m] = *(const bf16x8_t*)(img + abase[m] + aofs(0, d));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int R = 0; R < NROW; ++R) {
        const int ac = R & 1;
        if (R + 2 < NROW) {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) wrow[(R + 2) % 3][kx] = wload(wq, wlane, wofs(R + 2, kx));
        }
        if (R + 1 < NROW) {
#pragma unroll
            for (int d = 0; d < NSET; ++d)
#pragma unroll
                for (int m = 0; m < MB; ++m) a[ac ^ 1][d][m] = *(const bf16x8_t*)(img + abase[m] + aofs(R + 1, d));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int b = (kx + 1) & 1, d = (kx + 1) >> 1;            // kx 0 | 1 2 | 3 4 -> dx DXMAX | DXMAX-1 (x2) | DXMAX-2 (x2)
            const bf16x8_t w = __builtin_bit_cast(bf16x8_t, wrow[R % 3][kx]);
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a[ac][d][m], acc[m][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // masked store: this lane's pixel pair (iy, 2v | 2v+1), channels 32 cb + 8 g + 4 h + e = register 4 g + e = bit
    // 16 cb + 4 g + e of the pixel's mask word (dxo points at the clip)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int P = 32 * MB * wm + 32 * m + p31, ul = P / L::V, v = P - ul * L::V, iy = 2 * (u0 + ul) + A;
        if (P < G::ROWS * L::V && iy < G::HI) {
            const int pix = iy * G::WI + 2 * v;
            const bool two = EVEN || 2 * v + 1 < G::WI;
            const unsigned m0 = mw[m][0] >> (16 * cb), m1 = two ? mw[m][1] >> (16 * cb) : 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float2 o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    o[e].x = (m0 >> r) & 1u ? acc[m][0][r] : 0.f;
                    o[e].y = (m1 >> r) & 1u ? acc[m][1][r] : 0.f;
                    if constexpr (F32) {
                        float* q = dxo + (long)(32 * cb + 8 * g + 4 * h + e) * (G::HI * G::WI) + pix;
                        if (EVEN) *(float2*)q = o[e];
                        else { q[0] = o[e].x; if (two) q[1] = o[e].y; }
                    }
                    if constexpr (EVEN && G::OUT16) *(unsigned*)(dx16 + (long)(32 * cb + 8 * g + 4 * h + e) * (G::HI * G::WI) + pix) = pack_bf16(o[e].x, o[e].y);
                    bsum[r] += o[e].x + o[e].y;           // the bias gradient of the layer below: channel sums of dx
                }
                if constexpr (G::OUT8) {      // plane 4 cb + g of the C8 image, this lane's half of the two pixels' slots
                    uint2* q8 = dx8 + ((long)(4 * cb + g) * (G::HI * G::WI) + pix) * 2 + h;
                    q8[0] = make_uint2(pack_bf16(o[0].x, o[1].x), pack_bf16(o[2].x, o[3].x));
                    if (two) q8[2] = make_uint2(pack_bf16(o[0].y, o[1].y), pack_bf16(o[2].y, o[3].y));
                }
            }
        }
    }
}

template <class G, bool F32>
__global__ void __launch_bounds__(256) snd_dgrad_kernel(const uint4* __restrict__ gy8, const uint4* __restrict__ wp,
                                                        const unsigned* __restrict__ mask, float* __restrict__ dx, uint2* __restrict__ dx8,
                                                        unsigned short* __restrict__ dx16, float* __restrict__ bias_part, int nclips) {
    using L = DgLayout<G>;
    constexpr int MB = G::MB;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p31 = lane & 31;
    const int wm = wave & 1, cb = wave >> 1;
    float bsum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bsum[r] = 0.f;
    const int ntiles = nclips * G::TILES;
    int abase[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int P = 32 * MB * wm + 32 * m + p31, ul = P / L::V, v = P - ul * L::V;
        abase[m] = h * L::PLB + ul * L::SUBP + v * SLOT;
    }
    constexpr int NSL = L::NSL, NST = L::NST;
    uint4 sreg[NST];
    unsigned sok = 0u;
    auto stage_load = [&](int tile) {
        const int clip = tile / G::TILES, t = tile - clip * G::TILES, oy0 = G::ROWS * t + L::DYMIN;
        sok = 0u;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            int e = tid + 256 * k;
            asm volatile("" : "+v"(e));                     // keep the slots' index arithmetic out of the registers
            const int pl = e / (L::PR * G::WO), e2 = e - pl * (L::PR * G::WO), oy = oy0 + e2 / G::WO;
            const bool ok = e < NSL && (unsigned)oy < (unsigned)G::HO;
            sreg[k] = gy8[ok ? ((long)clip * 8 + pl) * (G::HO * G::WO) + oy0 * G::WO + e2 : 0];      // (zeroed at the LDS store)
            sok |= ok ? 1u << k : 0u;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            int e = tid + 256 * k;
            asm volatile("" : "+v"(e));
            const int pl = e / (L::PR * G::WO), e2 = e - pl * (L::PR * G::WO), i = e2 / G::WO;
            if (e < NSL)
                *(uint4*)(lds + buf * L::BUFB + pl * L::PLB + i * L::SUBP + (e2 - i * G::WO) * SLOT) = (sok >> k) & 1u ? sreg[k] : make_uint4(0, 0, 0, 0);
        }
    };
    // (rows past the patch and the slots past a row's WO pixels are only read by unused pixel slots / as the zero column)
    for (int i = tid; i < L::LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    int tile = blockIdx.x;
    if (tile < ntiles) { stage_load(tile); stage_store(0); }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t wq = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, NQ * L::NTAP * 2048, 0x00020000);
    const int wlane = cb * 1024 + lane * 16;
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int clip = tile / G::TILES, t = tile - clip * G::TILES, u0 = G::ROWS * t;
        const unsigned* mk = mask + (long)clip * (G::HI * G::WI) * 2;
        float* dxo = dx ? dx + (long)clip * CI * (G::HI * G::WI) : nullptr;
        uint2* dx8o = dx8 ? dx8 + (long)clip * 8 * (G::HI * G::WI) * 2 : nullptr;
        unsigned short* dx16o = dx16 ? dx16 + (long)clip * CI * (G::HI * G::WI) : nullptr;
        const int ntile = tile + (int)gridDim.x;
        const unsigned char* img = lds + buf * L::BUFB;
        dgrad_pass<G, 0, F32>(img, abase, wq, wlane, mk, cb, dxo, dx8o, dx16o, u0, wm, p31, h, bsum);
        if (ntile < ntiles) stage_load(ntile);
        __builtin_amdgcn_sched_barrier(0);
        dgrad_pass<G, 1, F32>(img, abase, wq, wlane, mk, cb, dxo, dx8o, dx16o, u0, wm, p31, h, bsum);
        if (ntile < ntiles) stage_store(buf ^ 1);
        __syncthreads();
    }
    // this workgroup's channel sums (fixed order: butterfly over the 32 pixel lanes, then the two pixel-half waves)
    float* red = (float*)lds;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = bsum[r];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d);
        if (p31 == 0) red[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = v;
    }
    __syncthreads();
    if (tid < 64) bias_part[blockIdx.x * 64 + tid] = red[(2 * (tid >> 5)) * 32 + (tid & 31)] + red[(2 * (tid >> 5) + 1) * 32 + (tid & 31)];
}

// gradient wrt conv 3's output: the (clip, oy, co*7 + ox) sequence, fp32 -> C8 bf16 (clip, 8, 73*7)
__global__ void __launch_bounds__(256) seq_to_c8_kernel(const float* __restrict__ g, uint4* __restrict__ y, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;              // (clip*8 + plane) * 511 + pix
    if (i >= total) return;
    const int pix = (int)(i % 511), oy = pix / 7, ox = pix - oy * 7;
    const long pl = i / 511, clip = pl >> 3;
    const float* src = g + (clip * 73 + oy) * 448 + (pl & 7) * 56 + ox;
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf16_bits(src[7 * j]);
    y[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// ---- weight gradient -------------------------------------------------------------------------------------------
// dW[co][ci][ky][kx] = sum_{n,oy,ox} gy[n][co][oy][ox] x[n][ci][2oy-PH+ky][2ox-PW+kx]: the MFMA's k index is the PIXEL,
// so both operands are read from their channel-innermost LDS images with the transposing ds_read_b64_tr_b16 (4 pixels
// x 16 channels per 16-lane group; each lane supplies its own row address, so the stride-2 walk through the x patch
// costs nothing).  A workgroup owns one 32-channel block of ci and 4 x TAPS taps for a group of clips; wave w owns TAPS
// taps and both 32-row blocks of co: 2 x TAPS accumulators that live across all the clips of the group and are written
// once, to the group's slab (tap, co, ci); a fold adds the slabs in fixed order.
// A k-step is RPK whole output rows (conv 2: one row of 13 pixels; conv 3: two rows of 7) padded to 16 slots whose gy
// is zero: 12-19 % of the matrix work is padding, but every operand address is a per-lane constant plus an immediate
// (with 16 consecutive pixels per k-step the row/column split of each pixel cost more instruction issue than the
// matrix instructions themselves -- one wave per SIMD hides nothing).  x patch 4 planes (parity-split as in the
// forward), gy 8 planes x 16 slots per k-step, double-buffered; every thread stages 20 fixed slots per tile, one pair
// per k-step, loaded a whole tile before they are stored.
struct WGeo2 {           // conv 2: 13 tiles of 12 output rows, 2 tap halves of 28
    static constexpr int HI = 300, WI = 20, HO = 150, WO = 13, KH = 11, KW = 5, PH = 5, PW = 5, NSLOT = 13;
    static constexpr int TR = 12, TILES = 13, RPK = 1, TAPS = 7, NTH = 2;
};
struct WGeo3 {           // conv 3: 4 tiles of 20 output rows, all 21 taps in one workgroup (6 per wave)
    static constexpr int HI = 150, WI = 13, HO = 73, WO = 7, KH = 7, KW = 3, PH = 1, PW = 1, NSLOT = 7;
    static constexpr int TR = 20, TILES = 4, RPK = 2, TAPS = 6, NTH = 1;
};
template <class G>
struct WgLayout {
    static constexpr int NTAP = G::KH * G::KW, KS = G::TR / G::RPK, XR = 2 * G::TR + G::KH - 2, SUBP = G::NSLOT * SLOT;
    static constexpr int XPARB = XR * SUBP;
    static constexpr int XPL = 2 * XPARB + (64 - (2 * XPARB) % 256 + 256) % 256;     // plane pitches = 64 mod 256: the four planes a
    static constexpr int GPL = KS * 256 + 64;                                        // half-wave reads fall on disjoint banks
    static constexpr int XB = 4 * XPL, BUFB = XB + 8 * GPL, LDSB = 2 * BUFB;
    static constexpr int XP = 256 / G::WI, NP = (XR + XP - 1) / XP;                  // x rows per staging pass, passes
    static constexpr int NPAIR = (4 * NP + 8 + 1) / 2;
    static_assert(XPL % 256 == 64 && GPL % 256 == 64 && LDSB <= 160 * 1024, "wgrad LDS");
    static_assert(G::TR % G::RPK == 0 && G::RPK * G::WO <= 16 && G::TR * G::WO <= 256 && NPAIR <= KS, "wgrad tile");
    static_assert(G::TILES * G::TR >= G::HO && 4 * G::TAPS * G::NTH >= NTAP, "wgrad cover");
};

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t tr_read2(const unsigned char* p0, const unsigned char* p1) {
    const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p0);
    const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <class G>
__global__ void __launch_bounds__(256) snd_wgrad_kernel(const uint4* __restrict__ x8, const uint4* __restrict__ gy8,
                                                        float* __restrict__ slab, int nclips, int per_group) {
    using L = WgLayout<G>;
    constexpr int TAPS = G::TAPS, KS = L::KS, NPAIR = L::NPAIR, NTAP = L::NTAP, SUBP = L::SUBP, NXS = 4 * L::NP;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int g16 = (lane >> 4) & 1, qp = (lane & 15) >> 2, p = lane & 3;
    const int cib = blockIdx.x & 1, th = (blockIdx.x >> 1) % G::NTH, grp = (blockIdx.x >> 1) / G::NTH;
    const int clip_lo = grp * per_group, clip_hi = min(nclips, clip_lo + per_group);
    const int ntiles = (clip_hi - clip_lo) * G::TILES;

    for (int i = tid; i < L::LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // per lane: plane / half-slot of its 4 channels and, for each of its two reads of a k-step, its pixel: slot
    // 8 h + qp (+ 4) of the k-step = (row within the k-step, column); plus this wave's taps (slots past the layer's last
    // tap repeat it and are not stored)
    const int chan = (2 * g16 + (p >> 1)), half8 = 8 * (p & 1);
    const int glane = L::XB + chan * L::GPL + half8 + (8 * h + qp) * SLOT;
    int xl[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int idx = 8 * h + 4 * rd + qp, rs = G::RPK == 1 ? 0 : idx / G::WO, ox = idx - rs * G::WO;
        xl[rd] = chan * L::XPL + half8 + rs * 2 * SUBP + ox * SLOT;
    }
    int tapbase[TAPS][2];
#pragma unroll
    for (int i = 0; i < TAPS; ++i) {
        const int t = min(th * 4 * TAPS + TAPS * wave + i, NTAP - 1), ky = t / G::KW, kx = t - ky * G::KW;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) tapbase[i][rd] = xl[rd] + (kx & 1) * L::XPARB + ky * SUBP + (kx >> 1) * SLOT;
    }

    // staging, fixed slots per thread: x = 4 planes x NP passes of XP rows x WI columns, gy = 8 planes x TR*WO pixels
    const int xr = tid / G::WI, xc = tid - xr * G::WI, xc5 = xc + G::PW;
    const bool xthr = tid < L::XP * G::WI, gthr = tid < G::TR * G::WO;
    const int xdst = (xc5 & 1) * L::XPARB + xr * SUBP + (xc5 >> 1) * SLOT;
    const int grow = tid / G::WO;
    const int gdst = L::XB + (grow / G::RPK) * 256 + ((grow % G::RPK) * G::WO + (tid - grow * G::WO)) * SLOT;
    // Every global load of the pipeline is UNCONDITIONAL -- clamped address, the zero for out-of-range slots chosen when the
    // slot is stored to LDS a tile later, the tile after next clamped to the last one instead of `if (tile + 2 < ntiles)` --:
    // with any load behind a branch hipcc cannot count the loads in flight and waits for (almost) all of them in the
    // middle of every tile, i.e. for loads issued a k-step earlier.
    auto slot_ok = [&](int tile, int k) -> bool {
        const int t = tile % G::TILES, oy0 = G::TR * t;
        if (k < NXS) {
            const int i = xr + L::XP * (k % L::NP), yy = 2 * oy0 - G::PH + i;
            return xthr && i < L::XR && (unsigned)yy < (unsigned)G::HI;
        }
        return k - NXS < 8 && gthr && oy0 + grow < G::HO;
    };
    auto slot_load = [&](int tile, int k) -> uint4 {
        const int clip = clip_lo + tile / G::TILES, t = tile % G::TILES, oy0 = G::TR * t;
        if (k < NXS) {
            const int pl = k / L::NP, i = xr + L::XP * (k % L::NP), yy = 2 * oy0 - G::PH + i;
            const int yc = min(max(yy, 0), G::HI - 1), xcc = min(xc, G::WI - 1);
            return x8[((long)clip * 8 + 4 * cib + pl) * (G::HI * G::WI) + yc * G::WI + xcc];
        }
        const int pl = min(k - NXS, 7);
        return gy8[((long)clip * 8 + pl) * (G::HO * G::WO) + min(oy0 * G::WO + tid, G::HO * G::WO - 1)];
    };
    auto slot_store = [&](int buf, int k, uint4 v, bool ok) {
        unsigned char* base = lds + buf * L::BUFB;
        if (!ok) v = make_uint4(0, 0, 0, 0);
        if (k < NXS) {
            const int pl = k / L::NP, i = xr + L::XP * (k % L::NP);
            if (xthr && i < L::XR) *(uint4*)(base + pl * L::XPL + xdst + L::XP * (k % L::NP) * SUBP) = v;
        } else if (k - NXS < 8 && gthr) {
            *(uint4*)(base + (k - NXS) * L::GPL + gdst) = v;
        }
    };

    f32x16_t acc[TAPS][2];
#pragma unroll
    for (int i = 0; i < TAPS; ++i)
#pragma unroll
        for (int cob = 0; cob < 2; ++cob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][cob][r] = 0.f;

    // Slot pair j of tile T+1 is stored (into the other buffer) at k-step j of tile T and its registers are re-loaded at
    // once with the pair of tile T+2: every load has a whole tile to land.
    uint4 st[NPAIR][2];
    if (ntiles == 0) return;                               // (never: the launcher sizes the groups so that each has a clip)
    const int last = ntiles - 1;
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) { st[j][0] = slot_load(0, 2 * j); st[j][1] = slot_load(0, 2 * j + 1); }      // (all in flight, then stored)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) { slot_store(0, 2 * j, st[j][0], slot_ok(0, 2 * j)); slot_store(0, 2 * j + 1, st[j][1], slot_ok(0, 2 * j + 1)); }
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) { st[j][0] = slot_load(min(1, last), 2 * j); st[j][1] = slot_load(min(1, last), 2 * j + 1); }
    __syncthreads();

#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const unsigned char* img = lds + buf * L::BUFB;
        const int t1 = min(tile + 1, last), t2 = min(tile + 2, last);      // (past the end: the last tile again, never used)
        bf16x8_t a[2][2], b[2][TAPS];
        // fragment f of k-step ks (output rows RPK ks ..: x rows 2 RPK ks + ky): f = 0, 1: gy for the two co blocks; 2 + i: x for tap i
        auto frag = [&](int ks, int set, int f) {
            if (f < 2) a[set][f] = tr_read2(img + glane + f * 4 * L::GPL + ks * 256, img + glane + f * 4 * L::GPL + ks * 256 + 64);
            else b[set][f - 2] = tr_read2(img + tapbase[f - 2][0] + ks * G::RPK * 2 * SUBP, img + tapbase[f - 2][1] + ks * G::RPK * 2 * SUBP);
        };
#pragma unroll
        for (int f = 0; f < TAPS + 2; ++f) frag(0, 0, f);
        __builtin_amdgcn_sched_barrier(0);
        // One wave per SIMD: whatever is not issued in the shadow of a matrix instruction is paid in full, and hipcc
        // neither interleaves the transposed reads by itself nor under sched_group_barrier -- so the order is pinned by
        // hand: per pair of matrix instructions one or two fragments of the next k-step, the staging behind the last ones.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int set = ks & 1;
#pragma unroll
            for (int i = 0; i < TAPS; ++i) {
                if (ks + 1 < KS) {
                    if (i < 2) { frag(ks + 1, set ^ 1, 2 * i); frag(ks + 1, set ^ 1, 2 * i + 1); }
                    else frag(ks + 1, set ^ 1, i + 2);
                }
                if (ks < NPAIR && i >= TAPS - 2) {
                    const int q = i - (TAPS - 2);
                    slot_store(buf ^ 1, 2 * ks + q, st[ks][q], slot_ok(t1, 2 * ks + q));
                    st[ks][q] = slot_load(t2, 2 * ks + q);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][0], b[set][i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][1], b[set][i], acc[i][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // slab[grp][tap][co][ci]: lanes walk ci
    float* out = slab + (long)grp * NTAP * CO * CI + 32 * cib + (lane & 31);
#pragma unroll
    for (int i = 0; i < TAPS; ++i) {
        const int t = th * 4 * TAPS + TAPS * wave + i;
        if (t < NTAP) {
#pragma unroll
            for (int cob = 0; cob < 2; ++cob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = 32 * cob + (r & 3) + 8 * (r >> 2) + 4 * h;
                    out[((long)t * CO + co) * CI] = acc[i][cob][r];
                }
        }
    }
}

// dW (OIHW) += sum over the groups' slabs (tap, co, ci), in group order
__global__ void __launch_bounds__(256) snd_wgrad_fold_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ngroups, int ntap) {
    const int i = blockIdx.x * 256 + threadIdx.x;              // (tap, co, ci)
    if (i >= ntap * CO * CI) return;
    float a = 0.f;
    for (int g = 0; g < ngroups; ++g) a += slab[(long)g * ntap * CO * CI + i];
    const int ci = i & 63, co = (i >> 6) & 63, t = i >> 12;
    dw[((long)co * CI + ci) * ntap + t] += a;
}

// ---- conv 1 (1 -> 64 channels, 11x11 stride 2 pad 5, (600,40) -> (300,20)) -------------------------------------------
// One input channel: the whole clip (600 x 40 features) sits in LDS as bf16 with its padding (610 rows of 52 elements),
// and the GEMM's k index is the TAP, ordered (ky, kx padded to 16): a lane's 8 consecutive k are 8 consecutive input
// columns of one row -- four aligned ds_read_b32 (the stride-2 start 2 ox + 8 h is even) -- the 5 padding taps of a
// row carry zero weights.  The 22 filter fragments stay in registers.  The store writes the layer's output three ways
// at once: fp32 NCHW, the C8 bf16 image conv 2's kernels read, and the sign words conv 2's data gradient stores
// through (in the gather-GEMM version this layer cost 0.53 ms + a 0.25 ms conversion pass).
constexpr int C1_H = 600, C1_W = 40, C1_HO = 300, C1_WO = 20, C1_PITCH = 104, C1_ROWS = C1_H + 10 + 2;
constexpr int C1_LDSB = C1_ROWS * C1_PITCH, C1_NBLK = (C1_HO * C1_WO + 31) / 32;
static_assert(C1_HO * C1_WO == HI * WI, "conv 1's output is conv 2's input");

// OIHW (64,1,11,11) -> [ky][cb][lane][8]: lane (r, h) holds W[32 cb + r][ky][kx = 8 h + j], zero for kx >= 11
__global__ void __launch_bounds__(256) pack_w1_kernel(const float* __restrict__ w, uint4* __restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 11 * 2 * 64) return;
    const int lane = i & 63, cb = (i >> 6) & 1, ky = i >> 7, co = 32 * cb + (lane & 31), kx0 = 8 * (lane >> 5);
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = kx0 + j < 11 ? bf16_bits(w[(co * 11 + ky) * 11 + kx0 + j]) : 0u;
    wp[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// clips 0 .. n0-1 from x0, n0 .. nclips-1 from x1 (the positive and the negative sounds of a batch)
template <bool F32>
__global__ void __launch_bounds__(256, 2) snd1_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int n0,
                                                          const uint4* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ y, uint2* __restrict__ y8, unsigned* __restrict__ ymask,
                                                          int nclips) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p31 = lane & 31;
    // pixel blocks per pass: the 22 filter fragments (88 registers) and the 32 biases leave room for two blocks' accumulators (64) at two
    // waves per SIMD -- with four (128) the kernel spilled 88-140 registers
    constexpr int MB = 2;
    for (int i = tid; i < C1_LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    u32x4_t wf[11][2];
#pragma unroll
    for (int ky = 0; ky < 11; ++ky)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wf[ky][cb] = *(const u32x4_t*)(wp + (ky * 2 + cb) * 64 + lane);
    float bv[2][16];                                            // this lane's 32 output channels' biases
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[cb][r] = bias[32 * cb + 8 * (r >> 2) + 4 * h + (r & 3)];
    for (int clip = blockIdx.x; clip < nclips; clip += gridDim.x) {
        __syncthreads();                                        // (zero fill | the previous clip's reads) before the image changes
        const float4* src = (const float4*)(clip < n0 ? x0 + (long)clip * (C1_H * C1_W) : x1 + (long)(clip - n0) * (C1_H * C1_W));
        constexpr int NV = (C1_H * C1_W / 4 + 255) / 256;      // 24 float4 per thread, in two batches of 12 in flight (beside the filter's 88 registers)
        static_assert(NV % 2 == 0, "two batches");
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float4 v[NV / 2];
#pragma unroll
            for (int k = 0; k < NV / 2; ++k) { const int i = tid + 256 * (k + hb * (NV / 2)); v[k] = src[min(i, C1_H * C1_W / 4 - 1)]; }
#pragma unroll
            for (int k = 0; k < NV / 2; ++k) {                  // row r, columns 4 c4 .. 4 c4 + 3 -> elements 5 + column of row r + 5
                const int i = tid + 256 * (k + hb * (NV / 2)), r = i / (C1_W / 4), c4 = i - r * (C1_W / 4);
                if (i < C1_H * C1_W / 4) {                      // (odd first element: a 2-byte, a 4-byte and a 2-byte store)
                    unsigned char* d = lds + (r + 5) * C1_PITCH + 2 * (5 + 4 * c4);
                    *(unsigned short*)d = (unsigned short)bf16_bits(v[k].x);
                    *(unsigned*)(d + 2) = bf16_bits(v[k].y) | (bf16_bits(v[k].z) << 16);
                    *(unsigned short*)(d + 6) = (unsigned short)bf16_bits(v[k].w);
                }
            }
        }
        __syncthreads();
        const long oclip = clip;
        for (int g = wave; g < (C1_NBLK + MB - 1) / MB; g += 4) {     // MB 32-pixel blocks at a time
            f32x16_t acc[MB][2];
            int base[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int P = min(32 * (MB * g + m) + p31, C1_HO * C1_WO - 1), oy = P / C1_WO, ox = P - oy * C1_WO;
                base[m] = 2 * oy * C1_PITCH + 4 * ox + 16 * h;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][cb][r] = 0.f;
            }
#pragma unroll
            for (int ky = 0; ky < 11; ++ky) {
                u32x4_t b[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const unsigned* q = (const unsigned*)(lds + base[m] + ky * C1_PITCH);
                    b[m].x = q[0]; b[m].y = q[1]; b[m].z = q[2]; b[m].w = q[3];
                }
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[m][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[ky][cb]),
                                                                             __builtin_bit_cast(bf16x8_t, b[m]), acc[m][cb], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int P = 32 * (MB * g + m) + p31;
                if (P < C1_HO * C1_WO) {
                    unsigned mk = 0u;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int co = 32 * cb + 8 * q + 4 * h + e;
                                v[e] = fmaxf(acc[m][cb][4 * q + e] + bv[cb][4 * q + e], 0.f);
                                if constexpr (F32) y[(oclip * CO + co) * (C1_HO * C1_WO) + P] = v[e];
                            }
                            const unsigned p01 = pack_bf16(v[0], v[1]), p23 = pack_bf16(v[2], v[3]);
                            y8[((oclip * 8 + 4 * cb + q) * (C1_HO * C1_WO) + P) * 2 + h] = make_uint2(p01, p23);
                            mk |= ((p01 & 0xffffu ? 1u : 0u) | (p01 >> 16 ? 2u : 0u) | (p23 & 0xffffu ? 4u : 0u) | (p23 >> 16 ? 8u : 0u))
                                  << (16 * cb + 4 * q);
                        }
                    ymask[(oclip * (C1_HO * C1_WO) + P) * 2 + h] = mk;
                }
            }
        }
    }
}

// conv 1's weight gradient: dW[co][ky][kx] = sum_{clip,oy,ox} gy[clip][co][oy][ox] x[clip][2oy-5+ky][2ox-5+kx].  k = pixel,
// 16 flat pixels per step (6000 = 375 x 16); rows = co (gy as bf16 NCHW: a lane's 8 pixels are 16 contiguous bytes, loaded
// straight from HBM, written in that form by conv 2's data-gradient kernel); columns = taps, 6 blocks of (two ky) x (kx
// padded to 16).  A tap's 8 pixels of a k half are 8 consecutive DWORDS of one row of the clip's LDS image (stride-2 elements:
// the needed half of each dword is picked with v_perm); a k half that starts at column 16 wraps into the next output row
// after 4 pixels (+128 bytes).  Wave = (co block, 3 column blocks): 3 accumulators, no cross-wave reduction; one slab
// (64 x 192) per workgroup, folded in workgroup order.
constexpr int C1_COLS = 192;
__global__ void __launch_bounds__(256, 2) snd1_wgrad_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int n0,
                                                            const unsigned short* __restrict__ gy16, float* __restrict__ slab, int nclips) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
    const int cb = wave & 1, jb = wave >> 1, kyl = r >> 4, kx = r & 15;
    const unsigned sel = (kx & 1) ? 0x07060302u : 0x05040100u;
    for (int i = tid; i < C1_LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    f32x16_t acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int clip = blockIdx.x; clip < nclips; clip += gridDim.x) {
        __syncthreads();
        const float4* src = (const float4*)(clip < n0 ? x0 + (long)clip * (C1_H * C1_W) : x1 + (long)(clip - n0) * (C1_H * C1_W));
        constexpr int NV = (C1_H * C1_W / 4 + 255) / 256;
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) { const int i = tid + 256 * k; v[k] = src[min(i, C1_H * C1_W / 4 - 1)]; }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k, rr = i / (C1_W / 4), c4 = i - rr * (C1_W / 4);
            if (i < C1_H * C1_W / 4) {
                unsigned char* d = lds + (rr + 5) * C1_PITCH + 2 * (5 + 4 * c4);
                *(unsigned short*)d = (unsigned short)bf16_bits(v[k].x);
                *(unsigned*)(d + 2) = bf16_bits(v[k].y) | (bf16_bits(v[k].z) << 16);
                *(unsigned short*)(d + 6) = (unsigned short)bf16_bits(v[k].w);
            }
        }
        __syncthreads();
        const uint4* ga = (const uint4*)(gy16 + ((long)clip * CO + 32 * cb + r) * (C1_HO * C1_WO)) + h;     // + 2 ks: pixels 16 ks + 8 h ..
        // gy comes straight from HBM: DEPTH k-steps of loads in flight per wave, unconditional (past the end the last step
        // is loaded again) so that hipcc counts them instead of waiting for all: depth 4 (conditional) 189 us, 8: 161 us,
        // 16: 175 us (spills).  What is left is the LDS side: 12 narrow reads per step and wave for 3 matrix instructions
        constexpr int NKS = C1_HO * C1_WO / 16, DEPTH = 8;
        uint4 ring[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) ring[d] = ga[2 * d];
#pragma unroll 1
        for (int ks0 = 0; ks0 < NKS; ks0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int ks = ks0 + d;
                {
                    const uint4 a = ring[d];
                    ring[d] = ga[2 * min(ks + DEPTH, NKS - 1)];
                    if (ks >= NKS) continue;                     // uniform: the tail of the last group
                    const int P0 = 16 * ks + 8 * h, oy = P0 / C1_WO, ox0 = P0 - oy * C1_WO;
                    const unsigned char* xb = lds + (2 * oy + kyl) * C1_PITCH + ((2 * ox0 + kx) >> 1) * 4;
                    const int hi = 16 + (ox0 == 16 ? 128 : 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const unsigned* q0 = (const unsigned*)(xb + (jb + 2 * i) * 2 * C1_PITCH);
                        const unsigned* q1 = (const unsigned*)(xb + (jb + 2 * i) * 2 * C1_PITCH + hi);
                        u32x4_t b;
                        b.x = __builtin_amdgcn_perm(q0[1], q0[0], sel); b.y = __builtin_amdgcn_perm(q0[3], q0[2], sel);
                        b.z = __builtin_amdgcn_perm(q1[1], q1[0], sel); b.w = __builtin_amdgcn_perm(q1[3], q1[2], sel);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                                         acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* out = slab + (long)blockIdx.x * CO * C1_COLS + r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) out[(32 * cb + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_COLS + 32 * (jb + 2 * i)] = acc[i][q];
}

// dW (64,1,11,11) += the workgroups' slabs (co, 12 ky x 16 kx), in workgroup order: 16 slab elements x 16 slab chains per block
__global__ void __launch_bounds__(256) snd1_wgrad_fold_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nslabs) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;          // i = co * 192 + ky * 16 + kx
    float a = 0.f;
    for (int s = g; s < nslabs; s += 16) a += slab[(long)s * CO * C1_COLS + i];
    __shared__ float red[16][17];
    red[g][threadIdx.x & 15] = a;
    __syncthreads();
    if (g == 0) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
        const int co = i / C1_COLS, t = i - co * C1_COLS, ky = t >> 4, kx = t & 15;
        if (ky < 11 && kx < 11) dw[co * 121 + ky * 11 + kx] += v;
    }
}

}  // namespace

// workspace of the bf16 sound kernels for up to mc clips (byte offsets, 256-aligned):
//   x8    conv 1's output as C8 bf16 (mc, 8, 300*20) + m1: its sign words      (written by to_c8_mask_kernel)
//   y8    conv 2's output as C8 bf16 (mc, 8, 150*13) + m2: its sign words      (written by conv 2's forward)
//   gy8   gradient wrt conv 2's output, C8 bf16; g38: gradient wrt conv 3's output (mc, 8, 73*7), C8 bf16
//   g116  gradient wrt conv 1's output, bf16 NCHW (written by conv 2's data gradient, read by conv 1's weight gradient)
//   wp1 / wp2 / wpt2 / wp3 / wpt3: fragment-ordered filters (forward / data gradient)
struct BfWs { long x8, m1, y8, m2, gy8, g38, g116, wp2, wpt2, wp3, wpt3, wp1, total; };
constexpr long kWp2Bytes = (long)NQ * 55 * 2 * 64 * 16, kWp3Bytes = (long)NQ * 21 * 2 * 64 * 16;
static BfWs bf_ws(int mc) {
    BfWs w{};
    long o = 0;
    auto take = [&](long n) { const long at = o; o += (n + 255) & ~255L; return at; };
    w.x8 = take((long)mc * CI * 300 * 20 * 2); w.m1 = take((long)mc * 300 * 20 * 8);
    w.y8 = take((long)mc * CI * 150 * 13 * 2); w.m2 = take((long)mc * 150 * 13 * 8);
    w.gy8 = take((long)mc * CO * 150 * 13 * 2); w.g38 = take((long)mc * CO * 73 * 7 * 2);
    w.wp2 = take(kWp2Bytes); w.wpt2 = take(kWp2Bytes); w.wp3 = take(kWp3Bytes); w.wpt3 = take(kWp3Bytes);
    w.wp1 = take(11 * 2 * 64 * 16);
    w.g116 = take((long)mc * CO * 300 * 20 * 2);                 // gradient wrt conv 1's output, bf16 NCHW (conv 1's weight gradient)
    w.total = o;
    return w;
}
long snd_bf16_workspace_bytes(int nclips) { return bf_ws(nclips).total; }
template <class T> static T* at(void* ws, long off) { return (T*)((char*)ws + off); }

// conv 1 forward for n0 clips of MFCC features x0 and n1 of x1 (each (n,1,600,40); either may be absent), written as clips
// 0 .. n0+n1-1 of the layer's output: y (fp32 NCHW, optional) + its C8 image and sign words in the workspace (what
// snd2_bf16_fwd reads)
int snd1_bf16_fwd(var_ctx* c, hipStream_t s, const float* x0, int n0, const float* x1, int n1, const float* w, const float* bias,
                  float* y, int maxclips, void* ws) {
    const BfWs o = bf_ws(maxclips);
    hipLaunchKernelGGL(pack_w1_kernel, dim3((11 * 2 * 64 + 255) / 256), dim3(256), 0, s, w, at<uint4>(ws, o.wp1));
    VAR_HIP_CHECK(c, hipGetLastError());
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd1_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDSB + 256));
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd1_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDSB + 256));
        attr |= var_dev_bit(c);
    }
    const int n = n0 + n1;
    if (y) hipLaunchKernelGGL(snd1_fwd_kernel<true>, dim3(n < 512 ? n : 512), dim3(256), C1_LDSB + 256, s, x0, x1, n0, at<uint4>(ws, o.wp1), bias,
                              y, at<uint2>(ws, o.x8), at<unsigned>(ws, o.m1), n);
    else hipLaunchKernelGGL(snd1_fwd_kernel<false>, dim3(n < 512 ? n : 512), dim3(256), C1_LDSB + 256, s, x0, x1, n0, at<uint4>(ws, o.wp1), bias,
                            y, at<uint2>(ws, o.x8), at<unsigned>(ws, o.m1), n);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// conv 2 forward: y (fp32 NCHW) from the C8 image of x in the workspace (snd1_bf16_fwd's by-product; or, x given, converted
// here from the fp32 map); also leaves y's C8 image + sign words for conv 3
int snd2_bf16_fwd(var_ctx* c, hipStream_t s, const float* x, const float* w, const float* bias, float* y, int nclips,
                  int maxclips, void* ws) {
    const BfWs o = bf_ws(maxclips);
    using L = FwdLayout<Geo2>;
    if (x) {
        const long total = (long)nclips * HI * WI;
        hipLaunchKernelGGL(to_c8_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, at<uint4>(ws, o.x8),
                           at<unsigned>(ws, o.m1), total, HI * WI);
        VAR_HIP_CHECK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(pack_w_kernel<false>, dim3((NQ * L::NTAP * 128 + 255) / 256), dim3(256), 0, s, w, at<uint4>(ws, o.wp2), L::NTAP);
    VAR_HIP_CHECK(c, hipGetLastError());
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_fwd_kernel<Geo2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDSB));
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_fwd_kernel<Geo2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDSB));
        attr |= var_dev_bit(c);
    }
    const int ntiles = nclips * Geo2::TILES;
    ProfScope prof(c, s, TAG_ITHOR_S2_FWD);
    if (y) hipLaunchKernelGGL((snd_fwd_kernel<Geo2, true>), dim3(ntiles < 256 ? ntiles : 256), dim3(256), L::LDSB, s, at<uint4>(ws, o.x8),
                              at<uint4>(ws, o.wp2), bias, y, at<uint2>(ws, o.y8), at<unsigned>(ws, o.m2), nclips);
    else hipLaunchKernelGGL((snd_fwd_kernel<Geo2, false>), dim3(ntiles < 256 ? ntiles : 256), dim3(256), L::LDSB, s, at<uint4>(ws, o.x8),
                            at<uint4>(ws, o.wp2), bias, y, at<uint2>(ws, o.y8), at<unsigned>(ws, o.m2), nclips);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// conv 3 forward: the GRU's input sequence y (clip, 73, 448), fp32, from conv 2's C8 image in the workspace
int snd3_bf16_fwd(var_ctx* c, hipStream_t s, const float* w, const float* bias, float* y, int nclips, int maxclips, void* ws) {
    const BfWs o = bf_ws(maxclips);
    using L = FwdLayout<Geo3>;
    hipLaunchKernelGGL(pack_w_kernel<false>, dim3((NQ * L::NTAP * 128 + 255) / 256), dim3(256), 0, s, w, at<uint4>(ws, o.wp3), L::NTAP);
    VAR_HIP_CHECK(c, hipGetLastError());
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_fwd_kernel<Geo3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDSB));
        attr |= var_dev_bit(c);
    }
    hipLaunchKernelGGL((snd_fwd_kernel<Geo3, true>), dim3(nclips < 256 ? nclips : 256), dim3(256), L::LDSB, s, at<uint4>(ws, o.y8),
                       at<uint4>(ws, o.wp3), bias, y, (uint2*)nullptr, (unsigned*)nullptr, nclips);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

template <class G>
static int dgrad_launch(var_ctx* c, hipStream_t s, const float* w, const uint4* gy8, uint4* wpt, const unsigned* mask, float* dx,
                        uint2* dx8, unsigned short* dx16, float* bias_part, int* nparts, int nclips, int tag) {
    using L = DgLayout<G>;
    hipLaunchKernelGGL(pack_w_kernel<true>, dim3((NQ * L::NTAP * 128 + 255) / 256), dim3(256), 0, s, w, wpt, L::NTAP);
    VAR_HIP_CHECK(c, hipGetLastError());
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_dgrad_kernel<G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDSB));
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_dgrad_kernel<G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDSB));
        attr |= var_dev_bit(c);
    }
    if ((dx8 != nullptr) != G::OUT8 || (dx16 != nullptr) != G::OUT16) { VAR_SET_ERR(c, "sound data gradient: output set does not match the layer"); return VAR_ERR_ARG; }
    const int ntiles = nclips * G::TILES;
    *nparts = ntiles < 256 ? ntiles : 256;
    ProfScope prof(c, s, tag);
    if (dx) hipLaunchKernelGGL((snd_dgrad_kernel<G, true>), dim3(*nparts), dim3(256), L::LDSB, s, gy8, wpt, mask, dx, dx8, dx16, bias_part, nclips);
    else hipLaunchKernelGGL((snd_dgrad_kernel<G, false>), dim3(*nparts), dim3(256), L::LDSB, s, gy8, wpt, mask, dx, dx8, dx16, bias_part, nclips);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// conv 2: dx (fp32 NCHW, masked by conv 1's sign words) from the gy image in the workspace
// bias_part: *nparts x 64 partial channel sums of dx (the bias gradient of the layer below)
int snd2_bf16_dgrad(var_ctx* c, hipStream_t s, const float* w, float* dx, float* bias_part, int* nparts, int nclips, int maxclips,
                    void* ws) {
    const BfWs o = bf_ws(maxclips);
    return dgrad_launch<DGeo2>(c, s, w, at<uint4>(ws, o.gy8), at<uint4>(ws, o.wpt2), at<unsigned>(ws, o.m1), dx, nullptr,
                               at<unsigned short>(ws, o.g116), bias_part, nparts, nclips, TAG_ITHOR_S2_DGRAD);
}

// conv 3: gy (the masked gradient of the GRU's input sequence, fp32) -> dx (fp32 NCHW, masked by conv 2's sign words), its C8
// image (= conv 2's gy image: no snd2_bf16_prepare_gy needed afterwards) and the partial channel sums of dx
int snd3_bf16_dgrad(var_ctx* c, hipStream_t s, const float* gy_seq, const float* w, float* dx, float* bias_part, int* nparts,
                    int nclips, int maxclips, void* ws) {
    const BfWs o = bf_ws(maxclips);
    const long total = (long)nclips * 8 * 511;
    hipLaunchKernelGGL(seq_to_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gy_seq, at<uint4>(ws, o.g38), total);
    VAR_HIP_CHECK(c, hipGetLastError());
    return dgrad_launch<DGeo3>(c, s, w, at<uint4>(ws, o.g38), at<uint4>(ws, o.wpt3), at<unsigned>(ws, o.m2), dx, at<uint2>(ws, o.gy8),
                               nullptr, bias_part, nparts, nclips, -1);
}

// gy (fp32 NCHW, already masked) -> its bf16 C8 image in the workspace: once per backward, before the two kernels that read it
int snd2_bf16_prepare_gy(var_ctx* c, hipStream_t s, const float* gy, int nclips, int maxclips, void* ws) {
    uint4* gy8 = at<uint4>(ws, bf_ws(maxclips).gy8);
    const long total = (long)nclips * 8 * HO * WO;
    hipLaunchKernelGGL(to_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gy, gy8, total, 8, HO * WO);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// dW += the weight gradient from the bf16 images in the workspace (x: the layer's input, gy: the gradient wrt its output);
// `slab` holds groups * NTAP*64*64 floats
template <class G>
static int wgrad_launch(var_ctx* c, hipStream_t s, const uint4* x8, const uint4* gy8, float* dw, float* slab, int nclips, int max_groups,
                        int tag) {
    using L = WgLayout<G>;
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd_wgrad_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDSB));
        attr |= var_dev_bit(c);
    }
    int groups = nclips < max_groups ? nclips : max_groups;
    const int per = (nclips + groups - 1) / groups;
    groups = (nclips + per - 1) / per;
    {
        ProfScope prof(c, s, tag);
        hipLaunchKernelGGL(snd_wgrad_kernel<G>, dim3(2 * G::NTH * groups), dim3(256), L::LDSB, s, x8, gy8, slab, nclips, per);
    }
    VAR_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(snd_wgrad_fold_kernel, dim3((L::NTAP * CO * CI + 255) / 256), dim3(256), 0, s, slab, dw, groups, L::NTAP);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
int snd2_bf16_wgrad(var_ctx* c, hipStream_t s, float* dw, float* slab, int nclips, int maxclips, void* ws) {
    const BfWs o = bf_ws(maxclips);
    return wgrad_launch<WGeo2>(c, s, at<uint4>(ws, o.x8), at<uint4>(ws, o.gy8), dw, slab, nclips, 64, TAG_ITHOR_S2_WGRAD);
}
// conv 3: x = conv 2's C8 image (its forward's by-product), gy = the C8 image of the sequence gradient -- call after snd3_bf16_dgrad
int snd3_bf16_wgrad(var_ctx* c, hipStream_t s, float* dw, float* slab, int nclips, int maxclips, void* ws) {
    const BfWs o = bf_ws(maxclips);
    return wgrad_launch<WGeo3>(c, s, at<uint4>(ws, o.y8), at<uint4>(ws, o.g38), dw, slab, nclips, 128, -1);
}

// conv 1's weight gradient: dw (64,1,11,11) += from the MFCC features (as snd1_bf16_fwd) and the bf16 gradient image conv 2's
// data gradient left in the workspace; slab: min(n0+n1, 512) x 64 x 192 floats
int snd1_bf16_wgrad(var_ctx* c, hipStream_t s, const float* x0, int n0, const float* x1, int n1, float* dw, float* slab, int maxclips,
                    void* ws) {
    const BfWs o = bf_ws(maxclips);
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)snd1_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDSB + 256));
        attr |= var_dev_bit(c);
    }
    const int n = n0 + n1, grid = n < 512 ? n : 512;          // (two co-resident workgroups per CU beat one with two clips: 183 vs 302 us)
    hipLaunchKernelGGL(snd1_wgrad_kernel, dim3(grid), dim3(256), C1_LDSB + 256, s, x0, x1, n0, at<unsigned short>(ws, o.g116), slab, n);
    VAR_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(snd1_wgrad_fold_kernel, dim3(CO * C1_COLS / 16), dim3(256), 0, s, slab, dw, grid);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
