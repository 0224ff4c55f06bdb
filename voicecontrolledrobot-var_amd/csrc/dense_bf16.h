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
    const int group = MODE == 2 ? mt * nt : mt;            // tiles of one K split | m tiles of one n tile
    const int nchunks = mt * nt * zt / group, padded = (nchunks + 7) / 8 * 8;
    const dim3 grid(padded * group);
    if (p.a16 && p.b16) hipLaunchKernelGGL((kernel<AK, BK, MODE, true, true>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    else if (p.b16) hipLaunchKernelGGL((kernel<AK, BK, MODE, false, true>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    else if (p.a16) { VAR_SET_ERR(c, "dense16: a bf16 copy of A alone is not instantiated"); return VAR_ERR_ARG; }
    else hipLaunchKernelGGL((kernel<AK, BK, MODE, false, false>), grid, dim3(256), 0, s, p, mt, nt, group, nchunks);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

}  // namespace dense16
