// Gather-GEMM engine of the iTHOR model (csrc/ithor.hip): one f32-MFMA kernel, C[m][n] = sum_k A(m,k) * B(k,n),
// whose two operands and whose store are POLICIES -- small structs that turn (m,k) / (k,n) / (m,n) into
// addresses.  Convolution forward, data gradient and weight gradient (any filter size, stride 1|2, padding), the
// Linear layers and every GRU product are instances of it; nothing is unfolded ("im2col") in HBM.
//
// Tile: 128 (m, the "lane" side: MFMA columns, contiguous in the stores) x NT (n, the "row" side, 32 or 64) per
// 256-thread workgroup, K in chunks of 16 through double-buffered LDS with the next chunk's global loads in
// flight during the MFMAs.  Wave w owns columns [32w, 32w+32) and all NT rows: NT/32 accumulators of
// v_mfma_f32_32x32x2f32.  A policy says for each operand whether consecutive lanes should walk m/n or k
// (whichever is contiguous in memory), and splits its index arithmetic into a per-m (or per-n) part hoisted out
// of the K loop and a per-k part.
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

template <class P, int NT>
__global__ void __launch_bounds__(256) gg_kernel(const P p) {
    constexpr int AS = GG_MT + 4, BS = NT + 4;
    constexpr int NB = NT / 32;
    __shared__ float As[2][GG_KC][AS];
    __shared__ float Bs[2][GG_KC][BS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * GG_MT, n0 = blockIdx.y * NT;
    // grid.z = batches x K splits
    const int nsplit = p.nsplit;
    const int bz = blockIdx.z / nsplit, sz = blockIdx.z - bz * nsplit;
    int kchunks = (p.K + GG_KC - 1) / GG_KC;
    const int per = (kchunks + nsplit - 1) / nsplit;
    const int c_lo = sz * per, c_hi = min(kchunks, c_lo + per);
    if (c_lo >= c_hi && nsplit > 1) return;

    // ---- loader mapping ----
    constexpr int NA = GG_MT * GG_KC / 256;                 // 8 elements of A per thread and chunk
    constexpr int NBE = NT * GG_KC / 256;                   // 4 | 2 elements of B
    typename P::AM am[P::A_KFAST ? NA : 1];
    typename P::BN bn[P::B_KFAST ? NBE : 1];
    if (P::A_KFAST) {
#pragma unroll
        for (int i = 0; i < NA; ++i) am[P::A_KFAST ? i : 0] = p.a_m(m0 + (tid >> 4) + 16 * i, bz);
    } else {
        am[0] = p.a_m(m0 + (tid & (GG_MT - 1)), bz);
    }
    if (P::B_KFAST) {
#pragma unroll
        for (int i = 0; i < NBE; ++i) bn[P::B_KFAST ? i : 0] = p.b_n(n0 + (tid >> 4) + 16 * i, bz);
    } else {
        bn[0] = p.b_n(n0 + (tid & (NT - 1)), bz);
    }

    float ra[NA], rb[NBE];
    auto gload = [&](int chunk) {
        const int k0 = chunk * GG_KC;
        if (P::A_KFAST) {
            const typename P::AK ak = p.a_k(k0 + (tid & 15), bz);
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = p.a(am[P::A_KFAST ? i : 0], ak);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = p.a(am[0], p.a_k(k0 + (tid >> 7) + 2 * i, bz));
        }
        if (P::B_KFAST) {
            const typename P::BK bk = p.b_k(k0 + (tid & 15), bz);
#pragma unroll
            for (int i = 0; i < NBE; ++i) rb[i] = p.b(bk, bn[P::B_KFAST ? i : 0]);
        } else {
#pragma unroll
            for (int i = 0; i < NBE; ++i) rb[i] = p.b(p.b_k(k0 + tid / NT + (256 / NT) * i, bz), bn[0]);
        }
    };
    auto lstore = [&](int buf) {
        if (P::A_KFAST) {
#pragma unroll
            for (int i = 0; i < NA; ++i) As[buf][tid & 15][(tid >> 4) + 16 * i] = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) As[buf][(tid >> 7) + 2 * i][tid & (GG_MT - 1)] = ra[i];
        }
        if (P::B_KFAST) {
#pragma unroll
            for (int i = 0; i < NBE; ++i) Bs[buf][tid & 15][(tid >> 4) + 16 * i] = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < NBE; ++i) Bs[buf][tid / NT + (256 / NT) * i][tid & (NT - 1)] = rb[i];
        }
    };

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    if (c_lo < c_hi) {
        gload(c_lo);
        lstore(0);
    }
    __syncthreads();
#pragma unroll 1
    for (int c = c_lo; c < c_hi; ++c) {
        const int cur = (c - c_lo) & 1;
        if (c + 1 < c_hi) gload(c + 1);
#pragma unroll
        for (int kk = 0; kk < GG_KC / 2; ++kk) {
            const float av = As[cur][2 * kk + half][32 * wave + l31];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Bs[cur][2 * kk + half][32 * b + l31], av, acc[b], 0, 0, 0);
        }
        if (c + 1 < c_hi) lstore(cur ^ 1);
        __syncthreads();
    }

    const int m = m0 + 32 * wave + l31;
    if (m < p.M) {
        const typename P::CM cm = p.c_m(m, bz);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N) p.store(cm, n, acc[b][r], bz);
            }
    }
}

template <class P>
static int gg_launch(var_ctx* c, hipStream_t s, const P& p, int batches = 1) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return VAR_OK;
    dim3 grid((p.M + GG_MT - 1) / GG_MT, 1, batches * p.nsplit);
    if (p.N <= 32) {
        grid.y = 1;
        hipLaunchKernelGGL((gg_kernel<P, 32>), grid, dim3(256), 0, s, p);
    } else {
        grid.y = (p.N + 63) / 64;
        hipLaunchKernelGGL((gg_kernel<P, 64>), grid, dim3(256), 0, s, p);
    }
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Convolution policies.  Geometry of the filter is compile-time, tensor sizes are run-time.
template <int KH_, int KW_, int SH_, int SW_, int PH_, int PW_>
struct Geo {
    static constexpr int KH = KH_, KW = KW_, SH = SH_, SW = SW_, PH = PH_, PW = PW_, KHW = KH_ * KW_;
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

// y = relu(conv(x, w) + bias).  m = (b, oy, ox), n = cout, k = (ci, ky, kx) -- the filter's own OIHW order, so
// B(k, n) = w[n*K + k].  SEQ: write (b, oy, n, ox) instead of NCHW, the layout the GRU reads as (b, t, 448).
template <class G, bool U8, bool SEQ>
struct ConvFwdP {
    static constexpr bool A_KFAST = false, B_KFAST = true;
    int M, N, K, nsplit;
    ConvDims d;
    const void* x; const float* w; const float* bias; float* y;
    struct AM { long xoff; int iy0, ix0; bool ok; };
    struct AK { int coff, ky, kx; bool ok; };
    struct BN { int woff; bool ok; };
    struct BK { int k; bool ok; };
    struct CM { long off; };
    __device__ AM a_m(int m, int) const {
        AM s; s.ok = m < M;
        int pix; const int b = fdiv(s.ok ? m : 0, d.HO * d.WO, d.inv_howo, pix);
        int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
        s.xoff = (long)b * d.xb; s.iy0 = oy * G::SH - G::PH; s.ix0 = ox * G::SW - G::PW;
        return s;
    }
    __device__ AK a_k(int k, int) const {
        AK s; s.ok = k < K;
        const int ci = k / G::KHW, r = k - ci * G::KHW;
        s.ky = r / G::KW; s.kx = r - s.ky * G::KW; s.coff = ci * d.H * d.W;
        return s;
    }
    __device__ float a(const AM& sm, const AK& sk) const {
        const int iy = sm.iy0 + sk.ky, ix = sm.ix0 + sk.kx;
        if (!(sm.ok && sk.ok) || (unsigned)iy >= (unsigned)d.H || (unsigned)ix >= (unsigned)d.W) return 0.f;
        const long o = sm.xoff + sk.coff + iy * d.W + ix;
        if (U8) return (float)((const uint8_t*)x)[o] / 255.f;      // dataset.py:67-68
        return ((const float*)x)[o];
    }
    __device__ BN b_n(int n, int) const { return BN{n * K, n < N}; }
    __device__ BK b_k(int k, int) const { return BK{k, k < K}; }
    __device__ float b(const BK& sk, const BN& sn) const { return (sk.ok && sn.ok) ? w[sn.woff + sk.k] : 0.f; }
    __device__ CM c_m(int m, int) const {
        int pix; const int b = fdiv(m, d.HO * d.WO, d.inv_howo, pix);
        if (SEQ) {
            int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
            return CM{((long)(b * d.HO + oy) * d.COUT) * d.WO + ox};
        }
        return CM{(long)b * d.COUT * d.HO * d.WO + pix};
    }
    __device__ void store(const CM& cm, int n, float v, int) const {
        v += bias[n];
        y[cm.off + (long)n * (SEQ ? d.WO : d.HO * d.WO)] = v > 0.f ? v : 0.f;
    }
};

// dx = conv_transpose(gy, w): m = (b, y, x) input pixel, n = ci, k = (co, ky, kx);
// A(m,k) = gy[b][co][(y+PH-ky)/SH][(x+PW-kx)/SW] where that divides and lies inside, B(k,n) = w[co][n][ky][kx].
// gy must already carry the ReLU mask of its layer.  SEQ: gy in (b, oy, co, ox) layout.
template <class G, bool SEQ>
struct ConvDgradP {
    static constexpr bool A_KFAST = false, B_KFAST = false;
    int M, N, K, nsplit;
    ConvDims d;
    const float* gy; const float* w; float* dx;
    const float* mask;       // optional: the (post-ReLU) activation dx belongs to; dx is zeroed where it is not positive
    struct AM { long goff; int ty0, tx0; bool ok; };
    struct AK { int co, ky, kx; bool ok; };
    struct BN { int noff; bool ok; };
    struct BK { int koff; bool ok; };
    struct CM { long off; };
    __device__ AM a_m(int m, int) const {
        AM s; s.ok = m < M;
        int pix; const int b = fdiv(s.ok ? m : 0, d.H * d.W, d.inv_hw, pix);
        int xx; const int yy = fdiv(pix, d.W, d.inv_w, xx);
        s.goff = (long)b * d.COUT * d.HO * d.WO; s.ty0 = yy + G::PH; s.tx0 = xx + G::PW;
        return s;
    }
    __device__ AK a_k(int k, int) const {
        AK s; s.ok = k < K;
        s.co = k / G::KHW; const int r = k - s.co * G::KHW;
        s.ky = r / G::KW; s.kx = r - s.ky * G::KW;
        return s;
    }
    __device__ float a(const AM& sm, const AK& sk) const {
        const int ty = sm.ty0 - sk.ky, tx = sm.tx0 - sk.kx;
        if (!(sm.ok && sk.ok) || ty < 0 || tx < 0) return 0.f;
        if (G::SH == 2 && (ty & 1)) return 0.f;
        if (G::SW == 2 && (tx & 1)) return 0.f;
        const int oy = G::SH == 2 ? ty >> 1 : ty, ox = G::SW == 2 ? tx >> 1 : tx;
        if (oy >= d.HO || ox >= d.WO) return 0.f;
        if (SEQ) return gy[sm.goff + ((long)oy * d.COUT + sk.co) * d.WO + ox];
        return gy[sm.goff + ((long)sk.co * d.HO + oy) * d.WO + ox];
    }
    __device__ BN b_n(int n, int) const { return BN{n * G::KHW, n < N}; }
    __device__ BK b_k(int k, int) const {
        const int co = k / G::KHW, r = k - co * G::KHW;
        return BK{co * d.CIN * G::KHW + r, k < K};
    }
    __device__ float b(const BK& sk, const BN& sn) const { return (sk.ok && sn.ok) ? w[sk.koff + sn.noff] : 0.f; }
    __device__ CM c_m(int m, int) const {
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
// m = (b, y', x') with y = 2y'+cy, x = 2x'+cx; k = (co, i, j) with ky = ry+2i, kx = rx+2j.
template <class G, bool SEQ>
struct ConvDgradS2P {
    static constexpr bool A_KFAST = false, B_KFAST = false;
    static constexpr int NKY = (G::KH + 1) / 2, NKX = (G::KW + 1) / 2, NTAP = NKY * NKX;
    int M, N, K, nsplit;
    ConvDims d;
    int H2, W2; float inv_h2w2, inv_w2;
    const float* gy; const float* w; float* dx;
    const float* mask;
    struct AM { long goff; int oy0, ox0; bool ok; };
    struct AK { int co, i, j; bool ok; };
    struct BN { int noff; bool ok; };
    struct BK { int koff; bool ok; };
    struct CM { long off; bool ok; };
    __device__ AM a_m(int m, int z) const {
        AM s; const int cy = z >> 1, cx = z & 1;
        int pix; const int b = fdiv(m < M ? m : 0, H2 * W2, inv_h2w2, pix);
        int xp; const int yp = fdiv(pix, W2, inv_w2, xp);
        s.ok = m < M && 2 * yp + cy < d.H && 2 * xp + cx < d.W;
        s.goff = (long)b * d.COUT * d.HO * d.WO;
        s.oy0 = yp + ((cy + G::PH - ((cy + G::PH) & 1)) >> 1);
        s.ox0 = xp + ((cx + G::PW - ((cx + G::PW) & 1)) >> 1);
        return s;
    }
    __device__ AK a_k(int k, int z) const {
        AK s; const int ry = ((z >> 1) + G::PH) & 1, rx = ((z & 1) + G::PW) & 1;
        s.co = k / NTAP; const int r = k - s.co * NTAP;
        s.i = r / NKX; s.j = r - s.i * NKX;
        s.ok = k < K && ry + 2 * s.i < G::KH && rx + 2 * s.j < G::KW;
        return s;
    }
    __device__ float a(const AM& sm, const AK& sk) const {
        const int oy = sm.oy0 - sk.i, ox = sm.ox0 - sk.j;
        if (!(sm.ok && sk.ok) || (unsigned)oy >= (unsigned)d.HO || (unsigned)ox >= (unsigned)d.WO) return 0.f;
        if (SEQ) return gy[sm.goff + ((long)oy * d.COUT + sk.co) * d.WO + ox];
        return gy[sm.goff + ((long)sk.co * d.HO + oy) * d.WO + ox];
    }
    __device__ BN b_n(int n, int) const { return BN{n * G::KHW, n < N}; }
    __device__ BK b_k(int k, int z) const {
        const int ry = ((z >> 1) + G::PH) & 1, rx = ((z & 1) + G::PW) & 1;
        const int co = k / NTAP, r = k - co * NTAP;
        const int i = r / NKX, j = r - i * NKX;
        const int ky = ry + 2 * i, kx = rx + 2 * j;
        return BK{co * d.CIN * G::KHW + ky * G::KW + kx, k < K && ky < G::KH && kx < G::KW};
    }
    __device__ float b(const BK& sk, const BN& sn) const { return (sk.ok && sn.ok) ? w[sk.koff + sn.noff] : 0.f; }
    __device__ CM c_m(int m, int z) const {
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
// m = j = (ci, ky, kx), n = co, k = (b, oy, ox); both operands are read along k (pixels).  K is split over
// grid.z and the partial sums are added with float atomics into a zeroed dw.
template <class G, bool U8, bool SEQ>
struct ConvWgradP {
    static constexpr bool A_KFAST = true, B_KFAST = true;
    int M, N, K, nsplit;
    ConvDims d;
    const void* x; const float* gy; float* dw;
    struct AM { int off, dy, dx; bool ok; };
    struct AK { long xoff; int iy0, ix0; bool ok; };
    struct BN { int noff; bool ok; };
    struct BK { long goff; bool ok; };
    struct CM { int j; };
    __device__ AM a_m(int j, int) const {
        AM s; s.ok = j < M;
        const int jj = s.ok ? j : 0;
        const int ci = jj / G::KHW, r = jj - ci * G::KHW;
        const int ky = r / G::KW, kx = r - ky * G::KW;
        s.dy = ky - G::PH; s.dx = kx - G::PW; s.off = ci * d.H * d.W + s.dy * d.W + s.dx;
        return s;
    }
    __device__ AK a_k(int k, int) const {
        AK s; s.ok = k < K;
        int pix; const int b = fdiv(s.ok ? k : 0, d.HO * d.WO, d.inv_howo, pix);
        int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
        s.iy0 = oy * G::SH; s.ix0 = ox * G::SW; s.xoff = (long)b * d.xb + s.iy0 * d.W + s.ix0;
        return s;
    }
    __device__ float a(const AM& sm, const AK& sk) const {
        const int iy = sk.iy0 + sm.dy, ix = sk.ix0 + sm.dx;
        if (!(sm.ok && sk.ok) || (unsigned)iy >= (unsigned)d.H || (unsigned)ix >= (unsigned)d.W) return 0.f;
        const long o = sk.xoff + sm.off;
        if (U8) return (float)((const uint8_t*)x)[o] / 255.f;
        return ((const float*)x)[o];
    }
    __device__ BN b_n(int n, int) const { return BN{SEQ ? n * d.WO : n * d.HO * d.WO, n < N}; }
    __device__ BK b_k(int k, int) const {
        BK s; s.ok = k < K;
        int pix; const int b = fdiv(s.ok ? k : 0, d.HO * d.WO, d.inv_howo, pix);
        if (SEQ) {
            int ox; const int oy = fdiv(pix, d.WO, d.inv_wo, ox);
            s.goff = ((long)(b * d.HO + oy) * d.COUT) * d.WO + ox;
        } else {
            s.goff = (long)b * d.COUT * d.HO * d.WO + pix;
        }
        return s;
    }
    __device__ float b(const BK& sk, const BN& sn) const { return (sk.ok && sn.ok) ? gy[sk.goff + sn.noff] : 0.f; }
    __device__ CM c_m(int j, int) const { return CM{j}; }
    __device__ void store(const CM& cm, int n, float v, int) const { atomicAdd(dw + (long)n * M + cm.j, v); }
};

// ------------------------------------------------------------------------------------------------------------
// Dense products with run-time strides: C[m*scm + n*scn] (=|+=) sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
// (+ bias[m]) (ReLU).  AKF/BKF: the operand is contiguous along k.  `z` (grid.z batch, e.g. GRU direction) moves
// every pointer by its batch stride.  MODE 0: store, 1: add to C (one owner per element), 2: atomic add (split K).
template <bool AKF, bool BKF, int MODE>
struct DenseP {
    static constexpr bool A_KFAST = AKF, B_KFAST = BKF;
    int M, N, K, nsplit;
    const float* A; long sam, sak, zA;
    const float* Bm; long sbk, sbn, zB;
    float* C; long scm, scn, zC;
    const float* bias; long zbias;
    int relu;
    struct AM { long off; bool ok; };
    struct AK { long off; bool ok; };
    struct BN { long off; bool ok; };
    struct BK { long off; bool ok; };
    struct CM { long off; int m; };
    __device__ AM a_m(int m, int z) const { return AM{z * zA + m * sam, m < M}; }
    __device__ AK a_k(int k, int) const { return AK{k * sak, k < K}; }
    __device__ float a(const AM& sm, const AK& sk) const { return (sm.ok && sk.ok) ? A[sm.off + sk.off] : 0.f; }
    __device__ BN b_n(int n, int z) const { return BN{z * zB + n * sbn, n < N}; }
    __device__ BK b_k(int k, int) const { return BK{k * sbk, k < K}; }
    __device__ float b(const BK& sk, const BN& sn) const { return (sk.ok && sn.ok) ? Bm[sk.off + sn.off] : 0.f; }
    __device__ CM c_m(int m, int z) const { return CM{z * zC + m * scm, m}; }
    __device__ void store(const CM& cm, int n, float v, int z) const {
        float* q = C + cm.off + n * scn;
        if (MODE == 2) { atomicAdd(q, v); return; }
        if (bias) v += bias[z * zbias + cm.m];
        if (MODE == 1) v += *q;
        if (relu) v = v > 0.f ? v : 0.f;
        *q = v;
    }
};
