// Dense products of the iTHOR model in its bf16 mode (GRU input projection, dX, dW_ih, dW_hh: 2.5 of the step's 15 ms on
// the gather-GEMM): C[m][n] = sum_k A(m,k) B(k,n) for the DenseP descriptions of gg.h (fp32 operands at run-time
// strides, batches, K splits into slabs), on v_mfma_f32_32x32x16_bf16.
//
// Tile 128 (m) x 128 (n) x 32 (k) per 256-thread workgroup, waves 2 x 2 with 64 x 64 each (4 accumulators).  Each
// operand tile is loaded with 16-byte loads along whichever index is contiguous in memory, rounded to bf16 and stored
// in one of two LDS images:
//   K-contiguous operand  ->  [index][32 k], 64-byte rows, 16-byte chunks XOR-swizzled by (index / 4) % 4; a fragment is
//                             one ds_read_b128 (conflict-free)
//   index-contiguous      ->  [32 k][128 index], 320-byte rows (= 64 mod 256); a fragment is two transposing
//                             ds_read_b64_tr_b16 (4 k x 16 indices per 16-lane group), conflict-free
// so no operand needs a transposed copy in HBM.  The accumulator's lanes walk m (MFMA columns): stores are contiguous
// for scm = 1, as in gg.h.  Double-buffered LDS (2 x 18-20 KB), several workgroups per CU hide the staging latency.
#pragma once
#include "gg.h"

typedef __bf16 d16_bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 d16_bf16x8_t __attribute__((ext_vector_type(8)));

namespace dense16 {

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int ROWK = 64;                       // bytes per index row of a K-contiguous image
constexpr int ROWT = 320;                      // bytes per k row of an index-contiguous image
constexpr int IMGK = 128 * ROWK, IMGT = TK * ROWT;

__device__ __forceinline__ unsigned bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return bf16_bits(a) | (bf16_bits(b) << 16); }

// t or zero, component by component (a select of whole float4 values goes through scratch memory in this hipcc)
__device__ __forceinline__ float4 keep(const float4 t, bool ok) {
    return make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
}

// one operand's staging registers and its two halves of the pipeline.  The source is fp32 (rounded here) or, b16, a bf16
// copy of the same array (same element strides): half the bytes, no conversion.
template <bool KC, bool B16>
struct Stage {
    float4 v[B16 ? 2 : 4];
    // src: element (idx, k) at base[idx * s_idx + k * s_k]; idx0/k0 = tile origin; lim / K = number of valid indices / k
    // (every load is made from an in-range address and its value selected afterwards: hipcc turns `ok ? *p : zero` into a
    //  load through a select of pointers, one of them a zero in scratch memory -- a flat load)
    __device__ __forceinline__ void load(const void* __restrict__ base_, long s_idx, long s_k, int idx0, int k0, int lim, int K,
                                         int tid) {
        if constexpr (B16) {
            const unsigned short* base = (const unsigned short*)base_;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i;
                if (KC) {
                    const int idx = idx0 + (q >> 2), k = k0 + 8 * (q & 3);
                    const bool ok = idx < lim && k < K;                         // K % 8 == 0
                    v[i] = keep(*(const float4*)(base + (long)min(idx, lim - 1) * s_idx + min(k, K - 8)), ok);
                } else {
                    const int k = k0 + (q >> 4), idx = idx0 + 8 * (q & 15);
                    const bool ok = idx < lim && k < K;                         // lim % 8 == 0
                    v[i] = keep(*(const float4*)(base + (long)min(k, K - 1) * s_k + min(idx, lim - 8)), ok);
                }
            }
        } else {
            const float* base = (const float*)base_;
            if (KC) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = tid + 256 * i, idx = idx0 + (q >> 2), k = k0 + 8 * (q & 3);
                    const float* src = base + (long)min(idx, lim - 1) * s_idx;
                    const float4 t0 = *(const float4*)(src + min(k, K - 4)), t1 = *(const float4*)(src + min(k + 4, K - 4));
                    v[2 * i] = keep(t0, idx < lim && k < K);                      // K % 4 == 0
                    v[2 * i + 1] = keep(t1, idx < lim && k + 4 < K);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = tid + 256 * i, k = k0 + (f >> 5), idx = idx0 + 4 * (f & 31);
                    const float4 t = *(const float4*)(base + (long)min(k, K - 1) * s_k + min(idx, lim - 4));
                    v[i] = keep(t, idx < lim && k < K);                          // lim % 4 == 0
                }
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* img, int tid) const {
        if constexpr (B16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i;
                if (KC) {
                    const int idx = q >> 2, c = q & 3;
                    *(float4*)(img + idx * ROWK + ((c ^ ((idx >> 2) & 3)) << 4)) = v[i];
                } else {
                    const int k = q >> 4, i8 = q & 15;
                    *(float4*)(img + k * ROWT + i8 * 16) = v[i];
                }
            }
        } else {
        if (KC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + 256 * i, idx = q >> 2, c = q & 3;
                const float4 a = v[2 * i], b = v[2 * i + 1];
                *(uint4*)(img + idx * ROWK + ((c ^ ((idx >> 2) & 3)) << 4)) =
                    make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, k = f >> 5, i4 = f & 31;
                *(uint2*)(img + k * ROWT + i4 * 8) = make_uint2(pack2(v[i].x, v[i].y), pack2(v[i].z, v[i].w));
            }
        }
        }
    }
};

// fragment of block `blk` (32 indices) for k-step ks (0 | 1) of the tile
template <bool KC>
__device__ __forceinline__ d16_bf16x8_t frag(const unsigned char* img, int blk, int ks, int lane) {
    if (KC) {
        const int idx = 32 * blk + (lane & 31), c = 2 * ks + (lane >> 5);
        return *(const d16_bf16x8_t*)(img + idx * ROWK + ((c ^ ((idx >> 2) & 3)) << 4));
    } else {
        const int g16 = (lane >> 4) & 1, qp = (lane & 15) >> 2, p = lane & 3, h = lane >> 5;
        const unsigned char* q = img + (16 * ks + 8 * h + qp) * ROWT + (32 * blk + 16 * g16 + 4 * p) * 2;
        const d16_bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) d16_bf16x4_t*)q);
        const d16_bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) d16_bf16x4_t*)(q + 4 * ROWT));
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool AK, bool BK, int MODE, bool A16, bool B16>
__global__ void __launch_bounds__(256) kernel(const DenseP<AK, BK, MODE> p, int mt, int nt, int group, int nchunks) {
    constexpr int IA = AK ? IMGK : IMGT, IB = BK ? IMGK : IMGT;
    __shared__ __align__(16) unsigned char lds[2 * (IA + IB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    // Work item j = (z, n tile, m tile), m fastest.  Workgroup L runs on XCD L % 8 (round-robin dispatch): items are handed
    // out so that each run of `group` consecutive items -- the tiles that read the same operand panel (one K split's tiles,
    // or the m tiles of one n tile) -- stays on ONE XCD and meets its panel in that XCD's L2.
    const int L = blockIdx.x, chunk = (L & 7) + 8 * ((L >> 3) / group), j = chunk * group + (L >> 3) % group;
    if (chunk >= nchunks) return;
    const int mi = j % mt, ni = (j / mt) % nt, zi = j / (mt * nt);
    const int m0 = mi * TM, n0 = ni * TN;
    const int nsplit = p.nsplit, bz = zi / nsplit, sz = zi - bz * nsplit;
    const int ktiles = (p.K + TK - 1) / TK, per = (ktiles + nsplit - 1) / nsplit;
    const int t_lo = sz * per, t_hi = min(ktiles, t_lo + per);
    const void* A = A16 ? (const void*)((const unsigned short*)p.A + bz * p.zA) : (const void*)(p.A + bz * p.zA);
    const void* B = B16 ? (const void*)((const unsigned short*)p.Bm + bz * p.zB) : (const void*)(p.Bm + bz * p.zB);
    // (idx stride, k stride) of each operand
    const long a_si = p.sam, a_sk = p.sak, b_si = p.sbn, b_sk = p.sbk;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    Stage<AK, A16> sa;
    Stage<BK, B16> sb;
    if (t_lo < t_hi) {
        sa.load(A, a_si, a_sk, m0, t_lo * TK, p.M, p.K, tid);
        sb.load(B, b_si, b_sk, n0, t_lo * TK, p.N, p.K, tid);
        sa.store(lds, tid);
        sb.store(lds + IA, tid);
    }
    __syncthreads();
    for (int t = t_lo; t < t_hi; ++t) {
        const int buf = (t - t_lo) & 1;
        const unsigned char* ia = lds + buf * (IA + IB);
        const unsigned char* ib = ia + IA;
        const bool more = t + 1 < t_hi;
        if (more) {
            sa.load(A, a_si, a_sk, m0, (t + 1) * TK, p.M, p.K, tid);
            sb.load(B, b_si, b_sk, n0, (t + 1) * TK, p.N, p.K, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            d16_bf16x8_t fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[i] = frag<AK>(ia, 2 * wm + i, ks, lane); fb[i] = frag<BK>(ib, 2 * wn + i, ks, lane); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)      // rows = n (MFMA A operand), columns = m (lanes)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[jj], fa[i], acc[i][jj], 0, 0, 0);
        }
        if (more) {
            unsigned char* na = lds + (buf ^ 1) * (IA + IB);
            sa.store(na, tid);
            sb.store(na + IA, tid);
        }
        __syncthreads();
    }
    // (a K split left without tiles -- the host trims nsplit for 16-wide chunks -- stores zeros: its slab is still summed)

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 64 * wm + 32 * i + (lane & 31);
        if (m >= p.M) continue;
        const auto cm = p.c_m(m, bz, sz);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wn + 32 * jj + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < p.N) p.store(cm, n, acc[i][jj][r], bz);
            }
    }
}

// The same tile with a register ring of RING_D stages for products with a long K loop and both operands as bf16 copies (a
// stage is 8 + 8 registers): the loads of a k-step are in flight for RING_D - 1 steps instead of one.  Every load and LDS
// store is unconditional (hipcc then counts the loads in flight instead of waiting for all of them); steps past the
// split's end load zeros (the k limit handed to Stage::load is the split's end), so the trip count is rounded up to a
// multiple of RING_D.
constexpr int RING_D = 4;
template <bool AK, bool BK, int MODE>
__global__ void __launch_bounds__(256) ring_kernel(const DenseP<AK, BK, MODE> p, int mt, int nt, int group, int nchunks) {
    constexpr int IA = AK ? IMGK : IMGT, IB = BK ? IMGK : IMGT;
    __shared__ __align__(16) unsigned char lds[2 * (IA + IB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int L = blockIdx.x, chunk = (L & 7) + 8 * ((L >> 3) / group), j = chunk * group + (L >> 3) % group;
    if (chunk >= nchunks) return;
    const int mi = j % mt, ni = (j / mt) % nt, zi = j / (mt * nt);
    const int m0 = mi * TM, n0 = ni * TN;
    const int nsplit = p.nsplit, bz = zi / nsplit, sz = zi - bz * nsplit;
    const int ktiles = (p.K + TK - 1) / TK, per = (ktiles + nsplit - 1) / nsplit;
    const int t_lo = sz * per, t_hi = min(ktiles, t_lo + per);
    const int kend = min(p.K, t_hi * TK);                   // loads at k >= kend give zeros
    const void* A = (const void*)((const unsigned short*)p.A + bz * p.zA);
    const void* B = (const void*)((const unsigned short*)p.Bm + bz * p.zB);
    const long a_si = p.sam, a_sk = p.sak, b_si = p.sbn, b_sk = p.sbk;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    Stage<AK, true> ra[RING_D];
    Stage<BK, true> rb[RING_D];
#pragma unroll
    for (int d = 0; d < RING_D - 1; ++d) {
        ra[d].load(A, a_si, a_sk, m0, (t_lo + d) * TK, p.M, kend, tid);
        rb[d].load(B, b_si, b_sk, n0, (t_lo + d) * TK, p.N, kend, tid);
    }
    ra[0].store(lds, tid);
    rb[0].store(lds + IA, tid);
    __syncthreads();
    for (int t0 = t_lo; t0 < t_hi; t0 += RING_D) {
#pragma unroll
        for (int d = 0; d < RING_D; ++d) {
            const int t = t0 + d;
            ra[(d + RING_D - 1) % RING_D].load(A, a_si, a_sk, m0, (t + RING_D - 1) * TK, p.M, kend, tid);
            rb[(d + RING_D - 1) % RING_D].load(B, b_si, b_sk, n0, (t + RING_D - 1) * TK, p.N, kend, tid);
            const unsigned char* ia = lds + (d & 1) * (IA + IB);       // (t - t_lo) & 1 == d & 1: RING_D is even
            const unsigned char* ib = ia + IA;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                d16_bf16x8_t fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) { fa[i] = frag<AK>(ia, 2 * wm + i, ks, lane); fb[i] = frag<BK>(ib, 2 * wn + i, ks, lane); }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[jj], fa[i], acc[i][jj], 0, 0, 0);
            }
            unsigned char* na = lds + ((d + 1) & 1) * (IA + IB);
            ra[(d + 1) % RING_D].store(na, tid);
            rb[(d + 1) % RING_D].store(na + IA, tid);
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 64 * wm + 32 * i + (lane & 31);
        if (m >= p.M) continue;
        const auto cm = p.c_m(m, bz, sz);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wn + 32 * jj + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < p.N) p.store(cm, n, acc[i][jj][r], bz);
            }
    }
}
static_assert(RING_D % 2 == 0, "the LDS buffer parity follows the ring position");

// ---- K-contiguous x K-contiguous products with a short K and a very long N (the GRU input projection: M 1536, K 448,
// N 37 376 rows of (clip, t), fp32 output of 459 MB) -------------------------------------------------------------------
// In the tile kernel above every 128 x 128 output tile loads both operands again (230 KB from L2 for 14.7 MFLOP) through a
// one-step prefetch, and the product ran at 13 % of the matrix peak / 1.5 TB/s of stores.  Here a workgroup keeps its
// 128-row panel of A (all of K: 115 KB) in LDS for its whole life and streams B: k-steps of 32 through a register ring of
// 14 stages (a whole tile of loads in flight) and a double-buffered 8 KB LDS image, flattened over the workgroup's n
// tiles, so that the load stream never drains between tiles.  One workgroup per CU.  293 -> 222 us for the input
// projection (2.1 TB/s of fp32 stores: the store of C is what is left).
constexpr int RES_KSTEPS = 14, RES_K = RES_KSTEPS * TK, RES_RING = 14;
constexpr int RES_ROW = RES_K * 2 + 16;                   // bytes per A row: = 144 mod 256, 16 consecutive rows hit 16 different 16-byte bank groups
constexpr int RES_LDS = 128 * RES_ROW + 2 * IMGK;

template <int MODE>
__global__ void __launch_bounds__(256) resident_a_kernel(const DenseP<true, true, MODE> p, int mt, int ngroups) {
    extern __shared__ __align__(16) unsigned char rlds[];
    unsigned char* const wimg = rlds;                      // [128 m][RES_K] bf16, row pitch RES_ROW
    unsigned char* const ximg = rlds + 128 * RES_ROW;      // 2 x IMGK
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int mi = blockIdx.x % mt, grp = (blockIdx.x / mt) % ngroups, bz = blockIdx.x / (mt * ngroups);
    const int m0 = mi * TM;
    const unsigned short* A = (const unsigned short*)p.A + bz * p.zA;
    const unsigned short* B = (const unsigned short*)p.Bm + bz * p.zB;
    // the A panel: 128 rows x 56 16-byte chunks
    constexpr int PER = 128 * (RES_K / 8) / 256;           // 28 chunks per thread, seven loads in flight at a time
    static_assert(128 * (RES_K / 8) % 256 == 0 && PER % 7 == 0, "A panel chunks");
    for (int b0 = 0; b0 < PER; b0 += 7) {
        float4 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int q = tid + 256 * (b0 + u), row = q / (RES_K / 8), ch = q - row * (RES_K / 8);
            v[u] = *(const float4*)(A + (long)min(m0 + row, p.M - 1) * p.sam + 8 * ch);
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int q = tid + 256 * (b0 + u), row = q / (RES_K / 8), ch = q - row * (RES_K / 8);
            *(float4*)(wimg + row * RES_ROW + ch * 16) = keep(v[u], m0 + row < p.M);
        }
    }
    const int nt = (p.N + TN - 1) / TN;
    const int ntiles = grp < nt ? (nt - grp + ngroups - 1) / ngroups : 0;      // n tiles grp, grp + ngroups, ...
    Stage<true, true> ring[RES_RING];
    // stream position g = tile * 14 + ks
    // (every load and LDS store of the stream is unconditional -- past the end the last tile is loaded again and never used --:
    //  with `if (g < total)` around them hipcc cannot count the loads in flight and waits for ALL of them at every step)
    if (ntiles == 0) return;
    auto issue = [&](Stage<true, true>& st, int tile, int ks) {
        st.load(B, p.sbn, 1, (grp + min(tile, ntiles - 1) * ngroups) * TN, ks * TK, p.N, p.K, tid);
    };
#pragma unroll
    for (int r = 0; r < RES_RING - 1; ++r) issue(ring[r], 0, r);
    ring[0].store(ximg, tid);
    __syncthreads();
    float bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 64 * wm + 32 * i + (lane & 31);
        bv[i] = (p.bias && m < p.M) ? p.bias[bz * p.zbias + m] : 0.f;
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < RES_KSTEPS; ++ks) {
            const int g = tile * RES_KSTEPS + ks;
            // the stage that was stored to LDS before this step is free: it takes the load RES_RING - 1 steps ahead
            issue(ring[(ks + RES_RING - 1) % RES_RING], tile + (ks + RES_RING - 1) / RES_KSTEPS, (ks + RES_RING - 1) % RES_KSTEPS);
            const unsigned char* xb = ximg + (g & 1) * IMGK;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                d16_bf16x8_t fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = *(const d16_bf16x8_t*)(wimg + (32 * (2 * wm + i) + (lane & 31)) * RES_ROW + ((2 * (2 * ks + k2) + (lane >> 5)) << 4));
                    fb[i] = frag<true>(xb, 2 * wn + i, k2, lane);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[jj], fa[i], acc[i][jj], 0, 0, 0);
            }
            ring[(ks + 1) % RES_RING].store(ximg + ((g + 1) & 1) * IMGK, tid);
            __syncthreads();
        }
        // The tile's 64 stores per thread go out in one block BEHIND the loads of the whole next tile (the ring is a tile
        // deep): loads and stores share one in-order counter, so a load issued after a store waits for that store -- and
        // the stores complete slowly.  Measured: ring of 7 stages 231 us, 14 stages 222 us; the stores spread over the
        // next tile's k-steps (5 per step from a copy of the accumulators) 261 / 252 us.
        const int n0 = (grp + tile * ngroups) * TN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* cm = p.C + bz * p.zC + m0 + 64 * wm + 32 * i + (lane & 31);      // scm == 1, M % 128 == 0
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + 64 * wn + 32 * jj + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (n < p.N) {
                        float v = acc[i][jj][r] + bv[i];
                        if (MODE == 1) v += cm[(long)n * p.scn];
                        if (p.relu) v = v > 0.f ? v : 0.f;
                        cm[(long)n * p.scn] = v;
                    }
                }
        }
    }
}

template <bool AK, bool BK, int MODE>
inline bool resident_a_eligible(const DenseP<AK, BK, MODE>& p) {
    return AK && BK && MODE != 2 && p.a16 && p.b16 && p.K == RES_K && p.nsplit == 1 && p.sak == 1 && p.sbk == 1 && p.scm == 1 &&
           p.M % TM == 0 && p.N >= 8 * TN;
}

// shapes the kernel takes: 16-byte aligned runs along each operand's contiguous index
template <bool AK, bool BK, int MODE>
inline bool eligible(const DenseP<AK, BK, MODE>& p) {
    if (p.M < 64 || p.N < 64 || p.scm != 1) return false;
    const auto ptr = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const auto ok = [](bool kc, bool b16, long s_i, long s_k, int lim, int K, long z) {
        const int g = b16 ? 8 : 4;                       // elements per 16 bytes
        if (z % g) return false;
        return kc ? (s_k == 1 && s_i % g == 0 && K % g == 0) : (s_i == 1 && s_k % g == 0 && lim % g == 0);
    };
    return ptr(p.A) && ptr(p.Bm) && ok(AK, p.a16 != 0, p.sam, p.sak, p.M, p.K, p.zA) && ok(BK, p.b16 != 0, p.sbn, p.sbk, p.N, p.K, p.zB);
}

template <bool AK, bool BK, int MODE>
inline int launch(var_ctx* c, hipStream_t s, const DenseP<AK, BK, MODE>& p, int batches) {
    const int mt = (p.M + TM - 1) / TM, nt = (p.N + TN - 1) / TN, zt = batches * p.nsplit;
    if constexpr (AK && BK && MODE != 2) {
        if (resident_a_eligible(p)) {                      // one workgroup per CU, each with its A panel and a share of the n tiles
            static unsigned attr[2] = {0, 0};              // bit d: set on device d
            if (!(attr[MODE] & var_dev_bit(c))) {
                VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)resident_a_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS));
                attr[MODE] |= var_dev_bit(c);
            }
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
            int ngroups = cus / (mt * batches);
            if (ngroups < 1) ngroups = 1;
            if (ngroups > nt) ngroups = nt;
            hipLaunchKernelGGL(resident_a_kernel<MODE>, dim3(mt * ngroups * batches), dim3(256), RES_LDS, s, p, mt, ngroups);
            VAR_HIP_CHECK(c, hipGetLastError());
            return VAR_OK;
        }
    }
    const int group = MODE == 2 ? mt * nt : mt;            // tiles of one K split | m tiles of one n tile
    const int nchunks = mt * nt * zt / group, padded = (nchunks + 7) / 8 * 8;
    const dim3 grid(padded * group);
    const int ksteps = ((p.K + TK - 1) / TK + p.nsplit - 1) / p.nsplit;
    // (measured: dX, K 1536, B k-fast: 175 + 115 -> 135 + 92 us; dW, K 4672 per split, both operands index-fast: 184 -> 195 us)
    if ((AK || BK) && p.a16 && p.b16 && ksteps >= 32) hipLaunchKernelGGL((ring_kernel<AK, BK, MODE>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    else if (p.a16 && p.b16) hipLaunchKernelGGL((kernel<AK, BK, MODE, true, true>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    else if (p.b16) hipLaunchKernelGGL((kernel<AK, BK, MODE, false, true>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    else if (p.a16) { VAR_SET_ERR(c, "dense16: a bf16 copy of A alone is not instantiated"); return VAR_ERR_ARG; }
    else hipLaunchKernelGGL((kernel<AK, BK, MODE, false, false>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

}  // namespace dense16
