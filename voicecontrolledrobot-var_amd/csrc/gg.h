// Gather-GEMM engine of the iTHOR model (csrc/ithor.hip): one f32-MFMA kernel, C[m][n] = sum_k A(m,k) * B(k,n),
// whose two operands and whose store are POLICIES -- small structs that turn (m,k) / (k,n) / (m,n) into
// addresses.  Convolution forward, data gradient and weight gradient (any filter size, stride 1|2, padding), the
// Linear layers and every GRU product are instances of it; nothing is unfolded ("im2col") in HBM.
//
// Tile: 128 (m, the "lane" side: MFMA columns, contiguous in the stores) x NT (n, the "row" side, 32 or 64) per
// 256-thread workgroup, K in chunks of 16 through double-buffered LDS with the next chunk's global loads in
// flight during the MFMAs.  Wave w owns columns [32w, 32w+32) and all NT rows: NT/32 accumulators of
// v_mfma_f32_32x32x2f32.  A policy says for each operand whether consecutive lanes should walk m/n or k
// (whichever is contiguous in memory), and gives its index arithmetic as a per-m (or per-n) part, computed once per
// thread, plus a per-k part, computed once per K chunk for the whole workgroup.
#pragma once
#include "var_common.h"

// q = k / d for 0 <= k < 2^24 through a float reciprocal and one correction step
__device__ __forceinline__ int fdiv(int k, int d, float inv, int& rem) {
    int q = (int)((float)k * inv);
    int r = k - q * d;
    if (r < 0) { q--; r += d; }
    else if (r >= d) { q++; r -= d; }
    rem = r;
    return q;
}

constexpr int GG_MT = 128;
constexpr int GG_KC = 16;

// ------------------------------------------------------------------------------------------------------------
// The kernel.  Both operands are SEPARABLE gathers
//     A(m,k) = ok ? srcA[ fA(m) + gA(k) ] : 0,   ok = one side's bit set contained in the other's
// (a convolution's address is base(pixel) + offset(channel, tap), and "tap (ky,kx) of pixel (y,x) lies inside the
// map" is bit ky of a row mask and bit 16+kx of a column mask of the pixel).  The k-side (offset, bits) pairs of the 16
// k of a chunk are computed once per chunk by one wave into a small LDS table, two chunks ahead; every other thread
// pays one LDS read, an AND, a compare and an add per element.  Offsets are 32-bit BYTE offsets from the operand's
// base pointer (scalar base + vector offset addressing, no 64-bit arithmetic per element; every operand is < 4 GB);
// the loads are branch-free (clamped to offset 0) and the zeroing select is deferred to the LDS store, so that the
// loads of chunk c+1 stay in flight across the MFMAs of chunk c.  The first version computed the channel/tap split
// and the bounds tests per element: ~320 VALU/branch instructions per 16 MFMAs, i.e. VALU-bound.
struct SepM { unsigned base; unsigned mask; };     // per m (or n): byte offset, validity mask
struct SepK { unsigned off; unsigned bits; };      // per k
constexpr unsigned SEP_OK = 1u << 30;         // "index in range": set in every valid mask and in all bits
constexpr unsigned SEP_BAD = 1u << 31;        // never in a mask: bits of an out-of-range k

typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));

// BF = true: the operands are rounded to bf16 (v_cvt_pk_bf16_f32, round to nearest even) on their way from the fp32 LDS
// tile into v_mfma_f32_32x32x16_bf16 (lane (r, h) supplies k = 8h..8h+7); accumulation stays fp32.  Same loader, same
// tiles: 2 MFMAs of 32 cycles per 16 k instead of 16 of 64 -- BASELINE config 4's stated precision.
template <class P, int NT, int KC, bool BF>
__global__ void __launch_bounds__(256) gg_kernel(const P p) {
    constexpr int AS = GG_MT + 4, BS = NT + 4;
    constexpr int NB = NT / 32;
    __shared__ float As[2][KC][AS];
    __shared__ float Bs[2][KC][BS];
    __shared__ SepK ktA[2][KC];
    __shared__ SepK ktB[2][KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * GG_MT, n0 = blockIdx.y * NT;
    const int nsplit = p.nsplit;
    const int bz = blockIdx.z / nsplit, sz = blockIdx.z - bz * nsplit;
    const int kchunks = (p.k_extent(bz) + KC - 1) / KC;
    const int per = (kchunks + nsplit - 1) / nsplit;
    const int c_lo = sz * per, c_hi = min(kchunks, c_lo + per);
    if (c_lo >= c_hi) { if (nsplit > 1 || kchunks == 0) return; }

    constexpr int NA = GG_MT * KC / 256;
    constexpr int NBE = NT * KC / 256;
    SepM am[P::A_KFAST ? NA : 1];
    SepM bn[P::B_KFAST ? NBE : 1];
    if (P::A_KFAST) {
#pragma unroll
        for (int i = 0; i < NA; ++i) am[P::A_KFAST ? i : 0] = p.a_m(m0 + tid / KC + (256 / KC) * i, bz);
    } else {
        am[0] = p.a_m(m0 + (tid & (GG_MT - 1)), bz);
    }
    if (P::B_KFAST) {
#pragma unroll
        for (int i = 0; i < NBE; ++i) bn[P::B_KFAST ? i : 0] = p.b_n(n0 + tid / KC + (256 / KC) * i, bz);
    } else {
        bn[0] = p.b_n(n0 + (tid & (NT - 1)), bz);
    }

    // k-side table of chunk `chunk`, computed by the KC first lanes of one wave (the waves take turns)
    auto ktable = [&](int chunk) {
        if (wave == (chunk & 3) && lane < KC) {
            ktA[chunk & 1][lane] = p.a_k(chunk * KC + lane, bz);
            ktB[chunk & 1][lane] = p.b_k(chunk * KC + lane, bz);
        }
    };
    float ra[NA], rb[NBE];
    unsigned oka = 0, okb = 0;                 // validity of the elements in flight, applied at the LDS store
    auto gload = [&](int chunk) {
        const SepK* ta = ktA[chunk & 1];
        const SepK* tb = ktB[chunk & 1];
        oka = 0; okb = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const SepM sm = am[P::A_KFAST ? i : 0];
            const SepK sk = P::A_KFAST ? ta[tid % KC] : ta[(tid >> 7) + 2 * i];
            const bool ok = P::A_K_IN_M ? (sm.mask & sk.bits) == sk.bits : (sm.mask & sk.bits) == sm.mask;
            ra[i] = p.a_load(ok ? sm.base + sk.off : 0u);
            oka |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < NBE; ++i) {
            const SepM sn = bn[P::B_KFAST ? i : 0];
            const SepK sk = P::B_KFAST ? tb[tid % KC] : tb[tid / NT + (256 / NT) * i];
            const bool ok = (sn.mask & sk.bits) == sk.bits;
            rb[i] = p.b_load(ok ? sn.base + sk.off : 0u);
            okb |= ok ? (1u << i) : 0u;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float v = (oka >> i) & 1u ? ra[i] : 0.f;
            if (P::A_KFAST) As[buf][tid % KC][tid / KC + (256 / KC) * i] = v;
            else As[buf][(tid >> 7) + 2 * i][tid & (GG_MT - 1)] = v;
        }
#pragma unroll
        for (int i = 0; i < NBE; ++i) {
            const float v = (okb >> i) & 1u ? rb[i] : 0.f;
            if (P::B_KFAST) Bs[buf][tid % KC][tid / KC + (256 / KC) * i] = v;
            else Bs[buf][tid / NT + (256 / NT) * i][tid & (NT - 1)] = v;
        }
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    ktable(c_lo);
    ktable(c_lo + 1);
    __syncthreads();
    if (c_lo < c_hi) {
        gload(c_lo);
        lstore(0);
    }
    __syncthreads();
#pragma unroll 1
    for (int c = c_lo; c < c_hi; ++c) {
        const int cur = (c - c_lo) & 1;
        if (c + 1 < c_hi) gload(c + 1);
        ktable(c + 2);                       // slot of chunk c, whose table was last read in the previous iteration
        if constexpr (BF) {
#pragma unroll
            for (int q = 0; q < KC / 16; ++q) {
                const int k0 = 16 * q + 8 * half;
                bf16x8_ av;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x2_ t = __builtin_convertvector(
                        f32x2_{As[cur][k0 + 2 * j][32 * wave + l31], As[cur][k0 + 2 * j + 1][32 * wave + l31]}, bf16x2_);
                    av[2 * j] = t[0]; av[2 * j + 1] = t[1];
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    bf16x8_ bv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x2_ t = __builtin_convertvector(
                            f32x2_{Bs[cur][k0 + 2 * j][32 * b + l31], Bs[cur][k0 + 2 * j + 1][32 * b + l31]}, bf16x2_);
                        bv[2 * j] = t[0]; bv[2 * j + 1] = t[1];
                    }
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[b], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float av = As[cur][2 * kk + half][32 * wave + l31];
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Bs[cur][2 * kk + half][32 * b + l31], av, acc[b], 0, 0, 0);
            }
        }
        if (c + 1 < c_hi) lstore(cur ^ 1);
        __syncthreads();
    }

    const int m = m0 + 32 * wave + l31;
    if (m < p.M) {
        const typename P::CM cm = p.c_m(m, bz, sz);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N) p.store(cm, n, acc[b][r], bz);
            }
    }
}

template <class P, int KC = GG_KC, bool BF = false>
static int gg_launch(var_ctx* c, hipStream_t s, const P& p, int batches = 1) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return VAR_OK;
    dim3 grid((p.M + GG_MT - 1) / GG_MT, 1, batches * p.nsplit);
    if (p.N <= 32) {
        hipLaunchKernelGGL((gg_kernel<P, 32, KC, BF>), grid, dim3(256), 0, s, p);
    } else {
        grid.y = (p.N + 63) / 64;
        hipLaunchKernelGGL((gg_kernel<P, 64, KC, BF>), grid, dim3(256), 0, s, p);
    }
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// Convolution policies.  Geometry of the filter is compile-time, tensor sizes are run-time.
template <int KH_, int KW_, int SH_, int SW_, int PH_, int PW_>
struct Geo {
    static constexpr int KH = KH_, KW = KW_, SH = SH_, SW = SW_, PH = PH_, PW = PW_, KHW = KH_ * KW_;
    static_assert(KH_ <= 15 && KW_ <= 13, "row taps in mask bits 0..14, column taps in bits 16..28");
};

struct ConvDims {
    int B, CIN, H, W, COUT, HO, WO;
    float inv_howo, inv_wo, inv_hw, inv_w;
    long xb;                 // elements between consecutive input images (>= CIN*H*W: 4-channel observations)
};

inline ConvDims conv_dims(int B, int CIN, int H, int W, int COUT, int KH, int KW, int SH, int SW, int PH, int PW) {
    ConvDims d{};
    d.B = B; d.CIN = CIN; d.H = H; d.W = W; d.COUT = COUT;
    d.HO = (H + 2 * PH - KH) / SH + 1;
    d.WO = (W + 2 * PW - KW) / SW + 1;
    d.inv_howo = 1.f / (float)(d.HO * d.WO); d.inv_wo = 1.f / (float)d.WO;
    d.inv_hw = 1.f / (float)(H * W); d.inv_w = 1.f / (float)W;
    d.xb = (long)CIN * H * W;
    return d;
}

// float at a 32-bit byte offset from a base pointer (scalar base + vector offset addressing)
__device__ __forceinline__ float ldf(const void* base, unsigned byte_off) {
    return *(const float*)((const char*)base + byte_off);
}

// bit t of the result: 0 <= start + t*step < limit, for t < n  (row / column masks of a pixel)
__device__ __forceinline__ unsigned tap_mask(int start, int step, int n, int limit) {
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < n; ++t) m |= ((unsigned)(start + t * step) < (unsigned)limit) ? (1u << t) : 0u;
    return m;
}

// y = relu(conv(x, w) + bias).  m = (b, oy, ox), n = cout, k = (ci, ky, kx) -- the filter's own OIHW order, so
// B(k, n) = w[n*K + k].  SEQ: write (b, oy, n, ox) instead of NCHW, the layout the GRU reads as (b, t, 448).
template <class G, bool U8, bool SEQ>
struct ConvFwdP {
    static constexpr bool A_KFAST = false, B_KFAST = true, A_K_IN_M = true;
    static constexpr unsigned ES = U8 ? 1u : 4u;
    int M, N, K, nsplit;
    ConvDims d;
    const void* x; const float* w; const float* bias; float* y;
    float* slab; long sstride;      // nsplit > 1: split s stores raw partial sums at slab + s*sstride (y's layout);
                                    // gg_finish adds them in order, then bias and ReLU (small-batch inference)
    struct CM { long off; };
    __device__ int k_extent(int) const { return K; }
    __device__ SepM a_m(int m, int) const {
        int pix; const int b = fdiv(m < M ? m : 0, d.HO * d.WO, d.inv_howo, pix);
        int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
        const int iy0 = oy * G::SH - G::PH, ix0 = ox * G::SW - G::PW;
        SepM s;
        s.base = (unsigned)((int)(b * d.xb) + iy0 * d.W + ix0) * ES;
        s.mask = m < M ? (SEP_OK | tap_mask(iy0, 1, G::KH, d.H) | (tap_mask(ix0, 1, G::KW, d.W) << 16)) : 0u;
        return s;
    }
    __device__ SepK a_k(int k, int) const {
        const int ci = k / G::KHW, r = k - ci * G::KHW;
        const int ky = r / G::KW, kx = r - ky * G::KW;
        return SepK{(unsigned)(ci * d.H * d.W + ky * d.W + kx) * ES, k < K ? (SEP_OK | (1u << ky) | (1u << (16 + kx))) : SEP_BAD};
    }
    __device__ float a_load(unsigned o) const {
        if (U8) return (float)*((const uint8_t*)x + o) / 255.f;      // dataset.py:67-68
        return ldf(x, o);
    }
    __device__ SepM b_n(int n, int) const { return SepM{(unsigned)(n * K) * 4u, n < N ? SEP_OK : 0u}; }
    __device__ SepK b_k(int k, int) const { return SepK{(unsigned)k * 4u, k < K ? SEP_OK : SEP_BAD}; }
    __device__ float b_load(unsigned o) const { return ldf(w, o); }
    __device__ CM c_m(int m, int, int sz) const {
        int pix; const int b = fdiv(m, d.HO * d.WO, d.inv_howo, pix);
        const long so = nsplit > 1 ? sz * sstride : 0;
        if (SEQ) {
            int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
            return CM{so + ((long)(b * d.HO + oy) * d.COUT) * d.WO + ox};
        }
        return CM{so + (long)b * d.COUT * d.HO * d.WO + pix};
    }
    __device__ void store(const CM& cm, int n, float v, int) const {
        const long o = cm.off + (long)n * (SEQ ? d.WO : d.HO * d.WO);
        if (nsplit > 1) { slab[o] = v; return; }
        v += bias[n];
        y[o] = v > 0.f ? v : 0.f;
    }
};

// dx = conv_transpose(gy, w) for stride 1: m = (b, y, x) input pixel, n = ci, k = (co, ky, kx);
// A(m,k) = gy[b][co][y+PH-ky][x+PW-kx] where that lies inside, B(k,n) = w[co][n][ky][kx].
// gy must already carry the ReLU mask of its layer; `mask` (optional) is the activation dx belongs to.
template <class G>
struct ConvDgradP {
    static_assert(G::SH == 1 && G::SW == 1, "stride 2 uses ConvDgradS2P");
    static constexpr bool A_KFAST = false, B_KFAST = false, A_K_IN_M = true;
    int M, N, K, nsplit;
    ConvDims d;
    const float* gy; const float* w; float* dx;
    const float* mask;
    struct CM { long off; };
    __device__ int k_extent(int) const { return K; }
    __device__ SepM a_m(int m, int) const {
        int pix; const int b = fdiv(m < M ? m : 0, d.H * d.W, d.inv_hw, pix);
        int xx; const int yy = fdiv(pix, d.W, d.inv_w, xx);
        const int ty0 = yy + G::PH, tx0 = xx + G::PW;
        SepM s;
        s.base = (unsigned)(b * d.COUT * d.HO * d.WO + ty0 * d.WO + tx0) * 4u;
        s.mask = m < M ? (SEP_OK | tap_mask(ty0, -1, G::KH, d.HO) | (tap_mask(tx0, -1, G::KW, d.WO) << 16)) : 0u;
        return s;
    }
    __device__ SepK a_k(int k, int) const {
        const int co = k / G::KHW, r = k - co * G::KHW;
        const int ky = r / G::KW, kx = r - ky * G::KW;
        return SepK{(unsigned)(co * d.HO * d.WO - ky * d.WO - kx) * 4u, k < K ? (SEP_OK | (1u << ky) | (1u << (16 + kx))) : SEP_BAD};
    }
    __device__ float a_load(unsigned o) const { return ldf(gy, o); }
    __device__ SepM b_n(int n, int) const { return SepM{(unsigned)(n * G::KHW) * 4u, n < N ? SEP_OK : 0u}; }
    __device__ SepK b_k(int k, int) const {
        const int co = k / G::KHW, r = k - co * G::KHW;
        return SepK{(unsigned)(co * d.CIN * G::KHW + r) * 4u, k < K ? SEP_OK : SEP_BAD};
    }
    __device__ float b_load(unsigned o) const { return ldf(w, o); }
    __device__ CM c_m(int m, int, int) const {
        int pix; const int b = fdiv(m, d.H * d.W, d.inv_hw, pix);
        return CM{(long)b * d.CIN * d.H * d.W + pix};
    }
    __device__ void store(const CM& cm, int n, float v, int) const {
        const long o = cm.off + (long)n * d.H * d.W;
        if (mask && !(mask[o] > 0.f)) v = 0.f;
        dx[o] = v;
    }
};

// The same for stride 2 in both directions, without the structural zeros: an input pixel only meets the filter taps
// of its own parity (ky = (y+PH) mod 2, kx likewise), so the pixels are processed in four parity classes (grid.z)
// whose K runs over (co, taps of that parity) -- a quarter of the products of the plain gather form.
// m = (b, y', x') with y = 2y'+cy, x = 2x'+cx; k = (co, i, j) with ky = ry+2i, kx = rx+2j, i < nky(class), j < nkx.
template <class G, bool SEQ>
struct ConvDgradS2P {
    static constexpr bool A_KFAST = false, B_KFAST = false, A_K_IN_M = true;
    static constexpr int NKY = (G::KH + 1) / 2, NKX = (G::KW + 1) / 2;
    int M, N, K, nsplit;     // K = COUT * NKY * NKX (the largest class); a class's own extent is k_extent(z)
    ConvDims d;
    int H2, W2; float inv_h2w2, inv_w2;
    const float* gy; const float* w; float* dx;
    const float* mask;
    struct CM { long off; bool ok; };
    __device__ static int ry_of(int z) { return ((z >> 1) + G::PH) & 1; }
    __device__ static int rx_of(int z) { return ((z & 1) + G::PW) & 1; }
    __device__ static int nky_of(int z) { return (G::KH - ry_of(z) + 1) / 2; }
    __device__ static int nkx_of(int z) { return (G::KW - rx_of(z) + 1) / 2; }
    __device__ int k_extent(int z) const { return d.COUT * nky_of(z) * nkx_of(z); }
    __device__ SepM a_m(int m, int z) const {
        const int cy = z >> 1, cx = z & 1;
        int pix; const int b = fdiv(m < M ? m : 0, H2 * W2, inv_h2w2, pix);
        int xp; const int yp = fdiv(pix, W2, inv_w2, xp);
        const bool ok = m < M && 2 * yp + cy < d.H && 2 * xp + cx < d.W;
        const int oy0 = yp + ((cy + G::PH - ry_of(z)) >> 1), ox0 = xp + ((cx + G::PW - rx_of(z)) >> 1);
        SepM s;
        s.base = (unsigned)(b * d.COUT * d.HO * d.WO + (SEQ ? oy0 * d.COUT * d.WO + ox0 : oy0 * d.WO + ox0)) * 4u;
        s.mask = ok ? (SEP_OK | tap_mask(oy0, -1, NKY, d.HO) | (tap_mask(ox0, -1, NKX, d.WO) << 16)) : 0u;
        return s;
    }
    __device__ void split(int k, int z, int& co, int& i, int& j) const {
        const int nkx = nkx_of(z), ntap = nky_of(z) * nkx;
        co = k / ntap; const int r = k - co * ntap;
        i = r / nkx; j = r - i * nkx;
    }
    __device__ SepK a_k(int k, int z) const {
        int co, i, j; split(k, z, co, i, j);
        const int off = SEQ ? co * d.WO - i * d.COUT * d.WO - j : co * d.HO * d.WO - i * d.WO - j;
        return SepK{(unsigned)off * 4u, k < k_extent(z) ? (SEP_OK | (1u << i) | (1u << (16 + j))) : SEP_BAD};
    }
    __device__ float a_load(unsigned o) const { return ldf(gy, o); }
    __device__ SepM b_n(int n, int) const { return SepM{(unsigned)(n * G::KHW) * 4u, n < N ? SEP_OK : 0u}; }
    __device__ SepK b_k(int k, int z) const {
        int co, i, j; split(k, z, co, i, j);
        return SepK{(unsigned)(co * d.CIN * G::KHW + (ry_of(z) + 2 * i) * G::KW + rx_of(z) + 2 * j) * 4u,
                    k < k_extent(z) ? SEP_OK : SEP_BAD};
    }
    __device__ float b_load(unsigned o) const { return ldf(w, o); }
    __device__ CM c_m(int m, int z, int) const {
        const int cy = z >> 1, cx = z & 1;
        int pix; const int b = fdiv(m, H2 * W2, inv_h2w2, pix);
        int xp; const int yp = fdiv(pix, W2, inv_w2, xp);
        const int y = 2 * yp + cy, x = 2 * xp + cx;
        return CM{(long)b * d.CIN * d.H * d.W + (long)y * d.W + x, y < d.H && x < d.W};
    }
    __device__ void store(const CM& cm, int n, float v, int) const {
        if (!cm.ok) return;
        const long o = cm.off + (long)n * d.H * d.W;
        if (mask && !(mask[o] > 0.f)) v = 0.f;
        dx[o] = v;
    }
};

// dw[co][ci][ky][kx] += sum_{b,oy,ox} gy[b][co][oy][ox] * x[b][ci][oy*SH+ky-PH][ox*SW+kx-PW]:
// m = j = (ci, ky, kx), n = co, k = (b, oy, ox); both operands are read along k (pixels).  Here the row/column masks
// belong to k (the output pixel) and the single tap bits to m, so the containment test runs the other way
// (A_K_IN_M = false).  K is split over grid.z: split s writes its partial sums into slab s (plain stores), a
// fixed-order slab_reduce adds them to dw -- no float atomics, bitwise reproducible.  Unsplit: dw += directly.
template <class G, bool U8, bool SEQ>
struct ConvWgradP {
    static constexpr bool A_KFAST = true, B_KFAST = true, A_K_IN_M = false;
    static constexpr unsigned ES = U8 ? 1u : 4u;
    int M, N, K, nsplit;
    ConvDims d;
    const void* x; const float* gy; float* dw;
    float* slab;             // nsplit slabs of M*N floats (used when nsplit > 1)
    struct CM { float* q; };
    __device__ int k_extent(int) const { return K; }
    __device__ SepM a_m(int j, int) const {
        const int jj = j < M ? j : 0;
        const int ci = jj / G::KHW, r = jj - ci * G::KHW;
        const int ky = r / G::KW, kx = r - ky * G::KW;
        return SepM{(unsigned)(ci * d.H * d.W + (ky - G::PH) * d.W + (kx - G::PW)) * ES,
                    j < M ? (SEP_OK | (1u << ky) | (1u << (16 + kx))) : SEP_BAD};
    }
    __device__ SepK a_k(int k, int) const {
        int pix; const int b = fdiv(k < K ? k : 0, d.HO * d.WO, d.inv_howo, pix);
        int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
        const int iy0 = oy * G::SH, ix0 = ox * G::SW;
        SepK s;
        s.off = (unsigned)((int)(b * d.xb) + iy0 * d.W + ix0) * ES;
        s.bits = k < K ? (SEP_OK | tap_mask(iy0 - G::PH, 1, G::KH, d.H) | (tap_mask(ix0 - G::PW, 1, G::KW, d.W) << 16)) : 0u;
        return s;
    }
    __device__ float a_load(unsigned o) const {
        if (U8) return (float)*((const uint8_t*)x + o) / 255.f;
        return ldf(x, o);
    }
    __device__ SepM b_n(int n, int) const { return SepM{(unsigned)(SEQ ? n * d.WO : n * d.HO * d.WO) * 4u, n < N ? SEP_OK : 0u}; }
    __device__ SepK b_k(int k, int) const {
        int pix; const int b = fdiv(k < K ? k : 0, d.HO * d.WO, d.inv_howo, pix);
        SepK s;
        if (SEQ) {
            int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
            s.off = (unsigned)(((b * d.HO + oy) * d.COUT) * d.WO + ox) * 4u;
        } else {
            s.off = (unsigned)(b * d.COUT * d.HO * d.WO + pix) * 4u;
        }
        s.bits = k < K ? SEP_OK : SEP_BAD;
        return s;
    }
    __device__ float b_load(unsigned o) const { return ldf(gy, o); }
    __device__ CM c_m(int j, int, int sz) const { return CM{(nsplit > 1 ? slab + (long)sz * M * N : dw) + j}; }
    __device__ void store(const CM& cm, int n, float v, int) const {
        float* q = cm.q + (long)n * M;
        *q = nsplit > 1 ? v : *q + v;
    }
};

// ------------------------------------------------------------------------------------------------------------
// Dense products with run-time strides: C[m*scm + n*scn] (=|+=) sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
// (+ bias[m]) (ReLU).  AKF/BKF: the operand is contiguous along k.  `z` (grid.z batch, e.g. GRU direction) moves
// every pointer by its batch stride.  MODE 0: store, 1: add to C (one owner per element), 2: K split over grid.z,
// split s stores its partial sums at C + s*sC (the consumer adds the nsplit slabs in fixed order; no atomics).
template <bool AKF, bool BKF, int MODE>
struct DenseP {
    static constexpr bool A_KFAST = AKF, B_KFAST = BKF, A_K_IN_M = true;
    int M, N, K, nsplit;
    const float* A; long sam, sak, zA;
    const float* Bm; long sbk, sbn, zB;
    float* C; long scm, scn, zC, sC;
    const float* bias; long zbias;
    int relu;
    int a16, b16;            // dense_bf16.h only: A / Bm point at a bf16 copy of the operand (same element strides)
    struct CM { long off; int m; };
    __device__ int k_extent(int) const { return K; }
    __device__ SepM a_m(int m, int z) const { return SepM{(unsigned)(z * zA + m * sam) * 4u, m < M ? SEP_OK : 0u}; }
    __device__ SepK a_k(int k, int) const { return SepK{(unsigned)(k * sak) * 4u, k < K ? SEP_OK : SEP_BAD}; }
    __device__ float a_load(unsigned o) const { return ldf(A, o); }
    __device__ SepM b_n(int n, int z) const { return SepM{(unsigned)(z * zB + n * sbn) * 4u, n < N ? SEP_OK : 0u}; }
    __device__ SepK b_k(int k, int) const { return SepK{(unsigned)(k * sbk) * 4u, k < K ? SEP_OK : SEP_BAD}; }
    __device__ float b_load(unsigned o) const { return ldf(Bm, o); }
    __device__ CM c_m(int m, int z, int sz) const { return CM{z * zC + m * scm + (MODE == 2 ? sz * sC : 0), m}; }
    __device__ void store(const CM& cm, int n, float v, int z) const {
        float* q = C + cm.off + n * scn;
        if (MODE == 2) { *q = v; return; }
        if (bias) v += bias[z * zbias + cm.m];
        if (MODE == 1) v += *q;
        if (relu) v = v > 0.f ? v : 0.f;
        *q = v;
    }
};

// out[i] = act(bias[(i / plane) % C] + sum_{s < nsplit} slabs[s*stride + i]): the tail of a split-K forward product
static __global__ void gg_finish_kernel(float* __restrict__ out, const float* __restrict__ slabs, long n, int nsplit,
                                        long stride, const float* __restrict__ bias, int C, int plane, int relu) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = bias ? bias[(i / plane) % C] : 0.f;
    // eight loads in flight, added in slab order (a rolled loop waits for every load: 72 slabs took 18 us)
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = slabs[(s + u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < nsplit; ++s) v += slabs[s * stride + i];
    out[i] = relu ? (v > 0.f ? v : 0.f) : v;
}

// K splits that bring a small product to about one workgroup per CU, each split at least 2 chunks, none empty
static inline int gg_small_split(int tiles, int K, long out_floats, long slab_floats) {
    const int kchunks = (K + GG_KC - 1) / GG_KC;
    int ns = (256 + tiles - 1) / tiles;
    if (ns > kchunks / 2) ns = kchunks / 2;
    if ((long)ns * out_floats > slab_floats) ns = (int)(slab_floats / out_floats);
    if (ns < 2) return 1;
    const int per = (kchunks + ns - 1) / ns;
    return (kchunks + per - 1) / per;
}
