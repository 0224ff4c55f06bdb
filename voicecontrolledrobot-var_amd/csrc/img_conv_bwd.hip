// Image CNN backward (autograd of models/pretext/arm_pretext_model.py:9-18 under
// loss.backward(), VAR/pretext_VAR.py:68), hand-written for gfx950 f32 matrix cores.
//
// Kernels per conv layer  y = relu(conv3x3_s2_p1(x, W) + b):
//   dgrad : gx = (W^T (*) gy) . (x > 0)      -- gradient wrt the previous layer's pre-activation
//           (gy is already masked by y > 0 by whoever produced it).  Stride-2 transposed conv
//           done as 2 x 2 parity classes: an input pixel (iy,ix) receives taps
//           ky in {1} (iy even) or {0,2} (iy odd), same for x, so each class is a dense
//           implicit GEMM  D[c][pixel] = sum_{tap,n} Wd[tap][n][c] * gy[n][pixel shifted].
//           A lane owns the horizontally adjacent pair (ix=2i, ix=2i+1) -> 8-byte stores.
//   wgrad / reduce : img_wgrad.hip (by default on the same stream, between the dgrad kernels; VAR_STREAMS bit 2
//                    moves them to a stream of their own).
#include <stdio.h>
#include <stdlib.h>

#include "img_stage.h"
#define VAR_WGRAD_DEVICE_ONLY
#include "img_wgrad.hip"             // WgCfg, img_wgrad_body, the W84_* / W96_* configurations (device code only)
#undef VAR_WGRAD_DEVICE_ONLY

// ------------------------------------------------------------------------------------------
// dgrad
// ------------------------------------------------------------------------------------------
// KC = 0: every wave runs both row-parity classes of its items (NWI waves).
// KC > 0: 3*KC wave groups split the reduction of ONE item per wave: {py=0 | py=1 row tap 0 | py=1 row tap 1}
//         (equal work) x KC chunks of the gy channels; partial tiles are folded through LDS in fixed order.
template <int CIN_, int COUT_, int H_, int RI_, int NU_, int NWI_, int KC_ = 0>
struct DgCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, RI = RI_, NU = NU_, NWI = NWI_, KC = KC_;
    static constexpr int KS = KC ? 3 * KC : 1, NW = NWI * KS;
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int NR = RI / 2 + 1;            // gy rows per unit
    static constexpr int POW = WO + 1;               // + zero column at ox = WO
    static constexpr int PLANE = NR * POW;
    static constexpr int UNIT = COUT * PLANE;
    static constexpr int NB = (H + RI - 1) / RI;     // bands per image
    static constexpr int WH = (W + 1) / 2;           // pixel pairs per row
    static constexpr int PPU = (RI / 2) * WH;        // pixel pairs per unit and row-parity class
    static constexpr int NPP = NU * PPU;
    static constexpr int NPB = (NPP + 31) / 32;
    static constexpr int CBLK = CIN / 32;
    static constexpr int ITEMS = NPB * CBLK;         // per class
    static constexpr int IPC = (ITEMS + NWI - 1) / NWI;
    static constexpr int NSLOT = KC ? 3 * KC - 2 : 0;            // parked partial tile pairs per item
    static constexpr int RED_FLOATS = NWI * NSLOT * 2048;
    static constexpr int LDS_FLOATS = ((NU * UNIT > RED_FLOATS ? NU * UNIT : RED_FLOATS) + 3) / 4 * 4;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(RI % 2 == 0, "band must hold whole row pairs");
    static_assert(KC == 0 || ITEMS == NWI, "K-split path: exactly one item per wave group");
    static_assert(KC == 0 || (COUT / 2) % (8 * KC) == 0, "gy channel chunks must be whole prefetch blocks");
};

template <class C>
__device__ __forceinline__ void img_dgrad_body(const float* __restrict__ gy, const float* __restrict__ wd,
                                               const float* __restrict__ x, float* __restrict__ gx, int B, int bx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int unit0 = bx * C::NU;

    // only the pad column (ox = WO) needs clearing; all data cells are written by the staging pass
    lds_zero_cols<NT>(lds, C::NU * C::COUT * C::NR, C::POW, C::WO, 1, tid);
#pragma unroll 1
    for (int u = 0; u < C::NU; ++u) {
        const int unit = unit0 + u;
        const bool uvalid = unit < total_units;
        const int b = uvalid ? unit / C::NB : 0, band = unit % C::NB;
        stage_y_band<C::COUT, C::HO, C::WO, C::NR, C::POW, C::PLANE, NT>(
            lds + u * C::UNIT, gy + (size_t)b * C::COUT * C::HO * C::WO, band * (C::RI / 2), uvalid, tid);
    }
    __syncthreads();

    const float* wl = wd + half * C::CIN + l31;        // + (tap*COUT + n)*CIN + cb*32
    if constexpr (C::KC > 0) {
        const int wv = wave % C::NWI, ks = wave / C::NWI;
        const int part = ks % 3, kc = ks / 3;
        const int py = part ? 1 : 0, ky = part == 0 ? 1 : (part == 1 ? 0 : 2), doy = part == 1 ? 1 : 0;
        const int pb = wv % C::NPB, cb = wv / C::NPB;
        int pp = pb * 32 + l31;
        const bool ppvalid = pp < C::NPP;
        if (!ppvalid) pp = 0;
        const int u = pp / C::PPU, q = pp - u * C::PPU;
        const int j = q / C::WH, i = q - j * C::WH;
        const int unit = unit0 + u;
        const int b = unit / C::NB, band = unit - b * C::NB;
        constexpr int U = 8;
        constexpr int NC = (C::COUT / 2) / C::KC;                // n-pair steps of a slice
        constexpr int NBK = NC / U;
        const int lb = u * C::UNIT + j * C::POW + i + half * C::PLANE + doy * C::POW + (2 * kc * NC) * C::PLANE;
        const float* wc = wl + cb * 32 + (size_t)((ky * 3) * C::COUT + 2 * kc * NC) * C::CIN;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // ReLU-mask operand of the epilogue (the previous layer's activation at this lane's pixel pair): loaded
        // now, by the waves that will run an epilogue, so that its HBM/L2 latency hides behind the matrix work
        const bool leader = kc == 0 && part < 2;
        const int iy_pre = band * C::RI + 2 * j + py;
        const bool st_ok = ppvalid && unit < total_units && iy_pre < C::H;
        const size_t o_pre = (size_t)(st_ok ? b : 0) * C::CIN * C::H * C::W + (size_t)(st_ok ? iy_pre : 0) * C::W + 2 * i;
        float xm0[16], xm1[16];
        // (the branch is wave-uniform and the loads inside unconditional -- clamped addresses for lanes without a pixel --:
        //  behind the per-lane `st_ok` they were waited for before the branch was left, i.e. before the matrix work)
        if (leader) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const size_t oc = o_pre + (size_t)c * C::H * C::W;
                if constexpr (C::W % 2 == 0) {
                    const float2 xv = *(const float2*)(x + oc);
                    xm0[r] = xv.x; xm1[r] = xv.y;
                } else {
                    // (unconditional, from a clamped address, and NOT selected here: a load behind a branch is waited for at
                    //  once, and so is one whose value feeds a select -- the epilogue only uses xm1 where 2 i + 1 < W)
                    xm0[r] = x[oc];
                    xm1[r] = x[oc + (2 * i + 1 < C::W ? 1 : 0)];
                }
            }
        }
        float wb[2][3][U];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int uu = 0; uu < U; ++uu) wb[0][kx][uu] = wc[(size_t)(kx * C::COUT + 2 * uu) * C::CIN];
#pragma unroll
        for (int blk = 0; blk < NBK; ++blk) {
            if (blk + 1 < NBK) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int uu = 0; uu < U; ++uu)
                        wb[(blk + 1) & 1][kx][uu] = wc[(size_t)(kx * C::COUT + 2 * ((blk + 1) * U + uu)) * C::CIN];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                const float b0 = lds[lb + 2 * (blk * U + uu) * C::PLANE];
                const float b1 = lds[lb + 2 * (blk * U + uu) * C::PLANE + 1];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][1][uu], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][2][uu], b0, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][0][uu], b1, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // fold: leaders are (part 0, kc 0) for py = 0 and (part 1, kc 0) for py = 1
        // slot of a non-leader among its item's parked tiles; py=0 members first
        const int slot = part == 0 ? kc - 1 : (C::KC - 1) + (part == 1 ? kc - 1 : C::KC - 1 + kc);
        __syncthreads();                                    // the staged gy bands are dead now
        if (!leader) {
            float* d = lds + (size_t)(wv * C::NSLOT + slot) * 2048;
#pragma unroll
            for (int r = 0; r < 16; ++r) { d[r * 64 + lane] = acc0[r]; d[1024 + r * 64 + lane] = acc1[r]; }
        }
        __syncthreads();
        if (!leader) return;
        {
            const int s0 = part == 0 ? 0 : C::KC - 1, s1 = part == 0 ? C::KC - 1 : C::NSLOT;
            for (int sl = s0; sl < s1; ++sl) {
                const float* d = lds + (size_t)(wv * C::NSLOT + sl) * 2048;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] += d[r * 64 + lane]; acc1[r] += d[1024 + r * 64 + lane]; }
            }
        }
        if (st_ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const size_t oc = o_pre + (size_t)c * C::H * C::W;
                if constexpr (C::W % 2 == 0) {
                    float2 g;
                    g.x = xm0[r] > 0.f ? acc0[r] : 0.f;
                    g.y = xm1[r] > 0.f ? acc1[r] : 0.f;
                    *(float2*)(gx + oc) = g;
                } else {
                    gx[oc] = xm0[r] > 0.f ? acc0[r] : 0.f;
                    if (2 * i + 1 < C::W) gx[oc + 1] = xm1[r] > 0.f ? acc1[r] : 0.f;
                }
            }
        }
        return;
    }
#pragma unroll 1
    for (int ci = 0; ci < C::IPC; ++ci) {
        const int idx = wave + C::NW * ci;
        if (idx >= C::ITEMS) break;
        const int pb = idx % C::NPB, cb = idx / C::NPB;
        int pp = pb * 32 + l31;
        const bool ppvalid = pp < C::NPP;
        if (!ppvalid) pp = 0;
        const int u = pp / C::PPU, q = pp - u * C::PPU;
        const int j = q / C::WH, i = q - j * C::WH;
        const int base = u * C::UNIT + j * C::POW + i + half * C::PLANE;
        const int unit = unit0 + u;
        const int b = unit / C::NB, band = unit - b * C::NB;
        const float* wc = wl + cb * 32;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            // row taps: py=0 -> ky=1 (doy 0);  py=1 -> ky=0 (doy 1), ky=2 (doy 0).
            // Filter values (L2-resident packed image) are prefetched one block of U n-pairs ahead.
            constexpr int U = 8;
            constexpr int BPT = (C::COUT / 2) / U;               // blocks per row tap
            const int NBK = (py ? 2 : 1) * BPT;
            float wb[2][3][U];
            {
                const int ky0 = py ? 0 : 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        wb[0][kx][u] = wc[(size_t)((ky0 * 3 + kx) * C::COUT + 2 * u) * C::CIN];
            }
#pragma unroll
            for (int blk = 0; blk < (py ? 2 : 1) * BPT; ++blk) {
                const int t = blk / BPT, nb0 = (blk % BPT) * U;
                const int doy = (py && !t) ? 1 : 0;
                if (blk + 1 < NBK) {
                    const int t1 = (blk + 1) / BPT, nb1 = ((blk + 1) % BPT) * U;
                    const int ky1 = py ? (t1 ? 2 : 0) : 1;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            wb[(blk + 1) & 1][kx][u] = wc[(size_t)((ky1 * 3 + kx) * C::COUT + 2 * (nb1 + u)) * C::CIN];
                }
                __builtin_amdgcn_sched_barrier(0);      // prefetch loads stay above this block's MFMAs
                const int lb = base + doy * C::POW;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float b0 = lds[lb + 2 * (nb0 + u) * C::PLANE];
                    const float b1 = lds[lb + 2 * (nb0 + u) * C::PLANE + 1];
                    // kx=1 -> px=0 (dox 0); kx=2 -> px=1 (dox 0); kx=0 -> px=1 (dox 1)
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][1][u], b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][2][u], b0, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][0][u], b1, acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: rows = channel c, col (lane&31) = pixel pair
            const int iy = band * C::RI + 2 * j + py;
            if (ppvalid && unit < total_units && iy < C::H) {
                const int ix = 2 * i;
                const size_t o = (size_t)b * C::CIN * C::H * C::W + (size_t)iy * C::W + ix;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const size_t oc = o + (size_t)c * C::H * C::W;
                    if constexpr (C::W % 2 == 0) {
                        const float2 xv = *(const float2*)(x + oc);
                        float2 g;
                        g.x = xv.x > 0.f ? acc0[r] : 0.f;
                        g.y = xv.y > 0.f ? acc1[r] : 0.f;
                        *(float2*)(gx + oc) = g;
                    } else {
                        gx[oc] = x[oc] > 0.f ? acc0[r] : 0.f;
                        if (ix + 1 < C::W) gx[oc + 1] = x[oc + 1] > 0.f ? acc1[r] : 0.f;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dgrad on 16 x 16 x 4 matrix tiles (v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k][j = l & 15], D col = l & 15,
// row = 4 (l >> 4) + r) for the 11 x 11 / 12 x 12 layer: 36 pixel pairs per row-parity class fill 75 % of three
// 16-pair tiles against 56 % of two 32-pair tiles, and (pair tile, 16-channel tile) gives exactly 12 items -- one per
// wave, each wave runs both parity classes of its item: no K split, no fold, one barrier in the whole kernel.
// ------------------------------------------------------------------------------------------
typedef float f32x4d __attribute__((ext_vector_type(4)));
template <int CIN_, int COUT_, int H_, int RI_>
struct Dg16Cfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, RI = RI_;
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int NR = RI / 2 + 1, POW = WO + 1, PLANE = NR * POW, UNIT = COUT * PLANE;
    static constexpr int NB = (H + RI - 1) / RI;
    static constexpr int WH = (W + 1) / 2, NPP = (RI / 2) * WH;      // pixel pairs per class
    static constexpr int NPT = (NPP + 15) / 16, NCT = CIN / 16, NW = NPT * NCT;
    static constexpr int LDS_BYTES = ((UNIT + 3) & ~3) * 4;
    static_assert(RI % 2 == 0 && NW == 12 && COUT % 32 == 0, "12 items = 12 waves");
};

template <class C>
__device__ __forceinline__ void img_dgrad16_body(const float* __restrict__ gy, const float* __restrict__ wd,
                                                 const float* __restrict__ x, float* __restrict__ gx, int B, int bx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, l15 = lane & 15;
    const int unit = bx;
    const int b = unit / C::NB, band = unit - b * C::NB;
    lds_zero_cols<NT>(lds, C::COUT * C::NR, C::POW, C::WO, 1, tid);
    stage_y_band<C::COUT, C::HO, C::WO, C::NR, C::POW, C::PLANE, NT>(lds, gy + (size_t)b * C::COUT * C::HO * C::WO,
                                                                    band * (C::RI / 2), true, tid);
    const int pt = wave % C::NPT, ct = wave / C::NPT;
    int pp = pt * 16 + l15;
    const bool ppvalid = pp < C::NPP;
    if (!ppvalid) pp = 0;
    const int j = pp / C::WH, i = pp - j * C::WH;
    // ReLU-mask operand of both parity classes, in flight during the matrix work
    float xm[2][4][2];
    const size_t o0 = (size_t)b * C::CIN * C::H * C::W + (size_t)(ct * 16 + 4 * kq) * C::H * C::W + 2 * i;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const int iy = band * C::RI + 2 * j + py;
        const bool ok = ppvalid && iy < C::H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // (unconditional loads from clamped addresses: `ok ? x[oc] : 0.f` is compiled as a branch around each load with a
            //  full wait behind it -- sixteen load latencies in a row)
            const size_t oc = o0 + (size_t)r * C::H * C::W + (size_t)(ok ? iy : 0) * C::W;
            xm[py][r][0] = x[oc];                                            // (raw: the epilogue uses them only where ok / 2 i + 1 < W;
            xm[py][r][1] = x[oc + (2 * i + 1 < C::W ? 1 : 0)];               //  a select here would wait for the loads before the matrix work)
        }
    }
    __syncthreads();
    const float* wl = wd + (size_t)kq * C::CIN + ct * 16 + l15;          // + (tap*COUT + 4 s) * CIN
    const int lb0 = j * C::POW + i + kq * C::PLANE;                       // + doy*POW + 4 s * PLANE
    constexpr int U = 8, SPT = C::COUT / 4, BPK = SPT / U;                // blocks per row tap
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        f32x4d acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int NBK = (py ? 2 : 1) * BPK;
        float wb[2][3][U];
        {
            const int ky0 = py ? 0 : 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int u = 0; u < U; ++u) wb[0][kx][u] = wl[(size_t)((ky0 * 3 + kx) * C::COUT + 4 * u) * C::CIN];
        }
#pragma unroll
        for (int blk = 0; blk < (py ? 2 : 1) * BPK; ++blk) {
            const int t = blk / BPK, s0 = (blk % BPK) * U;
            const int doy = (py && !t) ? 1 : 0;
            if (blk + 1 < NBK) {
                const int t1 = (blk + 1) / BPK, s1 = ((blk + 1) % BPK) * U;
                const int ky1 = py ? (t1 ? 2 : 0) : 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        wb[(blk + 1) & 1][kx][u] = wl[(size_t)((ky1 * 3 + kx) * C::COUT + 4 * (s1 + u)) * C::CIN];
            }
            __builtin_amdgcn_sched_barrier(0);
            const int lb = lb0 + doy * C::POW;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float b0 = lds[lb + 4 * (s0 + u) * C::PLANE];
                const float b1 = lds[lb + 4 * (s0 + u) * C::PLANE + 1];
                // kx=1 -> px=0 (dox 0); kx=2 -> px=1 (dox 0); kx=0 -> px=1 (dox 1)
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[blk & 1][1][u], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[blk & 1][2][u], b0, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[blk & 1][0][u], b1, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int iy = band * C::RI + 2 * j + py;
        if (ppvalid && iy < C::H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t oc = o0 + (size_t)r * C::H * C::W + (size_t)iy * C::W;
                gx[oc] = xm[py][r][0] > 0.f ? acc0[r] : 0.f;
                if (2 * i + 1 < C::W) gx[oc + 1] = xm[py][r][1] > 0.f ? acc1[r] : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// One launch for the two kernels of a layer that consume the same gy and do not depend on each other: the weight
// gradient (persistent split-K workgroups, blocks [0, Gw * NCOMBO)) and the data gradient (one tile per workgroup,
// the blocks after).  A replayed step costs about 5 us per kernel in launch, fill and drain whatever the kernel
// does; side by side in one grid the drain of the first overlaps the fill of the second as well.
// ------------------------------------------------------------------------------------------
template <class WC, class DC, bool D16>
__global__ void __launch_bounds__(768)
img_bwd_pair_kernel(const void* __restrict__ wx, long wbstride, const float* __restrict__ gy, float* __restrict__ slabs,
                    int Gw, const float* __restrict__ wd, const float* __restrict__ x, float* __restrict__ gx, int B) {
    static_assert(WC::NW * 64 == 768 && DC::NW * 64 == 768, "both halves run 12 waves");
    const int nw = Gw * WC::NCOMBO;
    const int id = blockIdx.x;
    if (id < nw) {
        img_wgrad_body<WC>(wx, wbstride, nullptr, gy, slabs, B, id % Gw, id / Gw, Gw);
    } else {
        if constexpr (D16) img_dgrad16_body<DC>(gy, wd, x, gx, B, id - nw);
        else img_dgrad_body<DC>(gy, wd, x, gx, B, id - nw);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
//                 CIN COUT  H  RI NU NW
using G96_3 = Dg16Cfg<64, 64, 12, 12>;
using D96_2p = DgCfg<32, 64, 24, 10, 2, 4, 1>;    // 12-wave form for the paired grid: 2 bands of 10 rows, 4 items x 3 slices
using D96_4 = DgCfg<64, 64, 6, 6, 3, 2, 2>;

// The weight gradients of conv 5, 4 and 3 in ONE grid (84 x 84): their inputs -- gact[5..3] left by img_chain_kernel, act[4..2]
// -- are all there before it starts, and none of the three fills the GPU by itself (128 split-K workgroups each).  Longest first.
template <class W2, class W3, class W4>
__global__ void __launch_bounds__(768)
img_wgrad345_kernel(const float* __restrict__ x2, const float* __restrict__ x3, const float* __restrict__ x4,
                    const float* __restrict__ gy3, const float* __restrict__ gy4, const float* __restrict__ gy5,
                    float* __restrict__ slabs2, float* __restrict__ slabs3, float* __restrict__ slabs4, int G2, int G3, int G4, int B) {
    static_assert(W2::NW * 64 == 768 && W3::NW * 64 == 768 && W4::NW * 64 == 768, "12 waves each");
    int id = blockIdx.x;
    const int n2 = G2 * W2::NCOMBO, n3 = G3 * W3::NCOMBO;
    if (id < n2) { img_wgrad_body<W2>(x2, 32L * 21 * 21, nullptr, gy3, slabs2, B, id % G2, id / G2, G2); return; }
    id -= n2;
    if (id < n3) { img_wgrad_body<W3>(x3, 64L * 11 * 11, nullptr, gy4, slabs3, B, id % G3, id / G3, G3); return; }
    id -= n3;
    img_wgrad_body<W4>(x4, 64L * 6 * 6, nullptr, gy5, slabs4, B, id % G4, id / G4, G4);
}

static int launch_wgrad345(var_ctx* c, hipStream_t s, int B) {
    using W2 = W84_2; using W3 = W84_3; using W4 = W84_4;
    ProfScope prof(c, s, TAG_IMG_WGRAD0 + 2);
    constexpr int LDS_BYTES = W2::LDS_BYTES > W3::LDS_BYTES ? (W2::LDS_BYTES > W4::LDS_BYTES ? W2::LDS_BYTES : W4::LDS_BYTES)
                                                           : (W3::LDS_BYTES > W4::LDS_BYTES ? W3::LDS_BYTES : W4::LDS_BYTES);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_wgrad345_kernel<W2, W3, W4>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    auto groups = [&](int layer, int nb, int nu) {
        const int need = (B * nb + nu - 1) / nu, gmax = img_wgrad_groups(layer);
        return need < gmax ? need : gmax;
    };
    const int G2 = groups(2, W2::NB, W2::NU), G3 = groups(3, W3::NB, W3::NU), G4 = groups(4, W4::NB, W4::NU);
    c->wg_groups[2] = G2; c->wg_groups[3] = G3; c->wg_groups[4] = G4;
    hipLaunchKernelGGL((img_wgrad345_kernel<W2, W3, W4>), dim3(G2 * W2::NCOMBO + G3 * W3::NCOMBO + G4 * W4::NCOMBO), dim3(768),
                       LDS_BYTES, s, c->act[2], c->act[3], c->act[4], c->gact[3], c->gact[4], c->gact[5],
                       c->slabs + img_slab_offset(2), c->slabs + img_slab_offset(3), c->slabs + img_slab_offset(4), G2, G3, G4, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// weight gradient of layer l and data gradient of layer l in one grid (84 x 84 inputs, layers 2..4)
template <class WC, class DC, bool D16>
static int launch_pair(var_ctx* c, hipStream_t s, int layer, const void* wx, long wbstride, const float* gy,
                       const float* wd, const float* x, float* gx, int B) {
    ProfScope prof(c, s, TAG_IMG_WGRAD0 + layer);
    constexpr int LDS_BYTES = WC::LDS_BYTES > DC::LDS_BYTES ? WC::LDS_BYTES : DC::LDS_BYTES;
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_bwd_pair_kernel<WC, DC, D16>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const int need = (B * WC::NB + WC::NU - 1) / WC::NU;
    const int gmax = img_wgrad_groups(layer);
    const int Gw = need < gmax ? need : gmax;
    c->wg_groups[layer] = Gw;
    int nd;
    if constexpr (D16) nd = B * DC::NB; else nd = (B * DC::NB + DC::NU - 1) / DC::NU;
    hipLaunchKernelGGL((img_bwd_pair_kernel<WC, DC, D16>), dim3(Gw * WC::NCOMBO + nd), dim3(768), LDS_BYTES, s, wx, wbstride, gy,
                       c->slabs + img_slab_offset(layer), Gw, wd, x, gx, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// The image backward on stream s: {wgrad l || dgrad l} in one grid for l = 4..2, {wgrad 1 || fused tail (dgrad 1 + wgrad 0)}
// in one grid at 84 x 84 (two launches at 96 x 96, whose tail runs 9 waves), then ONE fold of all five layers' slabs.
int launch_img_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B) {
    const PackLayout& K = c->kl;
    int rc;
    const int H = c->H;
    const long bs[5] = {c->saved_bstride, 32L * c->hs[1] * c->hs[1], 32L * c->hs[2] * c->hs[2],
                        64L * c->hs[3] * c->hs[3], 64L * c->hs[4] * c->hs[4]};
    const void* xin[5] = {c->saved_image, c->act[1], c->act[2], c->act[3], c->act[4]};
    if (H != 84 && H != 96) { VAR_SET_ERR(c, "unsupported image size %d (84 or 96)", H); return VAR_ERR_ARG; }
    if (H == 84) {
        // data gradients of conv 5 -> 4 -> 3 as one per-image chain (img_chain.hip), then their three weight gradients in one grid
        if ((rc = launch_img_bwd_chain(c, s, B)) != VAR_OK) return rc;
        if ((rc = launch_wgrad345(c, s, B)) != VAR_OK) return rc;
    }
    for (int l = 4; l >= 2 && H != 84; --l) {
        const float* gyl = c->gact[l + 1];
        const float* wdl = c->wpack + K.img_d[l];
        // (96 x 96: round 2's paired grids, weight gradient || data gradient of a layer)
        if (l == 4) rc = launch_pair<W96_4, D96_4, false>(c, s, 4, xin[4], bs[4], gyl, wdl, c->act[4], c->gact[4], B);
        else if (l == 3) rc = launch_pair<W96_3, G96_3, true>(c, s, 3, xin[3], bs[3], gyl, wdl, c->act[3], c->gact[3], B);
        else rc = launch_pair<W96_2, D96_2p, false>(c, s, 2, xin[2], bs[2], gyl, wdl, c->act[2], c->gact[2], B);
        if (rc != VAR_OK) return rc;
    }
    if (H == 84) {
        rc = launch_img_bwd_tail2(c, s, B);      // wgrad 1 + dgrad 1 + wgrad 0 in one role-specialised kernel (img_tail2.hip; act1 band-tiled)
    } else {
        if ((rc = launch_img_wgrad1_96(c, s, xin[1], bs[1], c->gact[2], B)) != VAR_OK) return rc;
        rc = launch_img_bwd_tail(c, s, B);
    }
    if (rc != VAR_OK) return rc;
    (void)params;
    return launch_img_wgrad_reduce(c, s, grads, 0, 4);
}
