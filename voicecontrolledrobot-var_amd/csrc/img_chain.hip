// Data gradients of conv 5 -> conv 4 -> conv 3 of the image CNN in ONE kernel, one workgroup per image (autograd of
// models/pretext/arm_pretext_model.py:13-18 under loss.backward(), VAR/pretext_VAR.py:68): the mirror image of the fused
// forward img_mid3.hip.  From the third conv on an image's gradients are small (gact5 2 KB, gact4 9 KB, gact3 31 KB):
// they stay in LDS between the layers (and also go to HBM, for the weight gradients), so the three dependent launches of
// round 2 -- each bound by its own per-image latency, 13 + 26 + 27 us alone -- become one chain without launch boundaries,
// restaging or index arithmetic per layer.
//
// Per layer  gx = (W^T (*) gy) . (x > 0):  stride-2 transposed conv as 2 x 2 parity classes, each a dense implicit GEMM
// D[c][pixel] on v_mfma_f32_16x16x4_f32 (16 channels x 16 pixels x 4 gy channels).  16 waves; a wave owns one work item
// = (parity class, 16-channel tile, up to four 16-pixel tiles): the filter slice of its class is streamed ONCE from the packed
// image in L2 (one 256-byte load per k-step, eight steps ahead) and used for all of the item's pixel tiles; the gy operand
// comes from the layer's LDS tile (zero row / column for the +1 shifts of the odd classes); the ReLU mask operand (the
// previous layer's activation) is fetched before the matrix work.  No K split, no fold.
#include <stdlib.h>

#include "var_common.h"

namespace {
PH_DECL();
}
#ifdef VAR_PHASES
extern "C" int var_debug_phases_chain(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
namespace {
typedef float f32x4c __attribute__((ext_vector_type(4)));

// gy tile of a layer in LDS: [n][HO + 1 rows][HO + 1 cols], last row / column zero
template <int COUT_, int CIN_, int H_>
struct ChL {
    static constexpr int COUT = COUT_, CIN = CIN_, H = H_;
    static constexpr int HO = (H - 1) / 2 + 1, POW = HO + 1, PLANE = POW * POW, FLOATS = COUT * PLANE;
    static constexpr int NCT = CIN / 16;
};
using L5 = ChL<64, 64, 6>;      // gy = gact5 (3 x 3)   -> gact4 (6 x 6)
using L4 = ChL<64, 64, 11>;     // gy = gact4 (6 x 6)   -> gact3 (11 x 11)
using L3 = ChL<64, 32, 21>;     // gy = gact3 (11 x 11) -> gact2 (21 x 21)
constexpr int kG5 = 0, kG4 = kG5 + ((L5::FLOATS + 3) & ~3), kG3 = kG4 + ((L4::FLOATS + 3) & ~3);
constexpr int kChainTileFloats = kG3 + ((L3::FLOATS + 3) & ~3);
constexpr int kW3 = kChainTileFloats, kW3Floats = 9 * L3::COUT * L3::CIN;       // conv 3's filter (A-fragment order), by LDS-DMA
constexpr int kChainLdsFloats = kW3 + kW3Floats;
static_assert(kChainLdsFloats * 4 <= 160 * 1024 && kW3 % 4 == 0 && kW3Floats % 256 == 0, "LDS budget, 16-byte DMA pieces");
constexpr int kChainNT = 1024;

// one work item: NT pixel tiles [t0, t0 + NT) of parity class cls, channel tile ct
template <class L, int NT>
__device__ __forceinline__ void chain_item(const float* __restrict__ wa, const float* __restrict__ gys, const float* __restrict__ xact,
                                           float* __restrict__ gx, float* __restrict__ nxt, int nxt_pow, int nxt_plane,
                                           int cls, int ct, int t0, int lane) {
    const int q = lane >> 4, l15 = lane & 15;
    const int py = cls >> 1, px = cls & 1;
    const int NI = (L::H + 1 - px) / 2, NJ = (L::H + 1 - py) / 2, npx = NI * NJ;
    int pj[NT], pi[NT];
    bool ok[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        int m = (t0 + u) * 16 + l15;
        ok[u] = m < npx;
        if (!ok[u]) m = 0;
        pj[u] = m / NI; pi[u] = m - pj[u] * NI;
    }
    // ReLU-mask operand (x > 0 at this lane's pixel, channels 16 ct + 4 q + r), in flight during the matrix work
    float mv[NT][4];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mv[u][r] = xact[((16 * ct + 4 * q + r) * L::H + 2 * pj[u] + py) * L::H + 2 * pi[u] + px];
    f32x4c acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[u] = {0.f, 0.f, 0.f, 0.f};
    const int ntx = px ? 2 : 1, ntap = (py ? 2 : 1) * ntx;
    constexpr int NSG = L::COUT / 16;                          // groups of 4 k-steps per tap
    static_assert(NSG == 4, "the ring below walks four groups per tap");
    // A operands: the transposed filter in A-fragment order (var_common.h: img_a), [tap][c tile][group][lane][4 k-steps] -- one
    // 16-byte load per lane and group.  From L2 (ALDS = false: three groups ahead) or from the copy LDS-DMA left in LDS.
    const float* wl = wa + (size_t)ct * NSG * 256 + 4 * lane;
    int lb[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) lb[u] = q * L::PLANE + pj[u] * L::POW + pi[u];
    auto tap_of = [&](int tt, int& tap, int& off) {
        const int ty = tt / ntx, tx = tt - ty * ntx;
        const int ky = py ? (ty ? 2 : 0) : 1, kx = px ? (tx ? 2 : 0) : 1;
        tap = ky * 3 + kx;
        off = (ky == 0 ? L::POW : 0) + (kx == 0 ? 1 : 0);     // gy row j + 1 for ky = 0, column i + 1 for kx = 0
    };
    auto a_of = [&](int tt, int sg) {
        int tap, off;
        tap_of(tt < ntap ? tt : ntap - 1, tap, off);
        return *(const f32x4c*)(wl + (size_t)(tap * L::NCT * NSG + sg) * 256);
    };
    f32x4c ar[4];                                              // ring: group g of tap tt lives in ar[g] (4 groups per tap)
    ar[0] = a_of(0, 0); ar[1] = a_of(0, 1); ar[2] = a_of(0, 2);
#pragma unroll 1
    for (int tt = 0; tt < ntap; ++tt) {
        int tap, off;
        tap_of(tt, tap, off);
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) {
            // three groups ahead: group sg + 3 of this tap, or group sg - 1 of the next
            ar[(sg + 3) & 3] = sg == 0 ? a_of(tt, 3) : a_of(tt + 1, sg - 1);
            // the group's gy operands are all read before its first MFMA (pinned: hipcc otherwise sinks each read to its use)
            float bq[4][NT];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < NT; ++u) bq[s][u] = gys[lb[u] + off + 4 * (4 * sg + s) * L::PLANE];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[sg][s], bq[s][u], acc[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: mask, HBM (NCHW, the weight gradients read it) and the next layer's LDS tile
    // (opaque pixel coordinates: hipcc otherwise forms the 4 NT 64-bit store addresses before the matrix loop and carries them through it)
#pragma unroll
    for (int u = 0; u < NT; ++u) { asm volatile("" : "+v"(pj[u])); asm volatile("" : "+v"(pi[u])); }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (!ok[u]) continue;
        const int y = 2 * pj[u] + py, x = 2 * pi[u] + px;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * ct + 4 * q + r;
            const float v = mv[u][r] > 0.f ? acc[u][r] : 0.f;
            gx[(c * L::H + y) * L::H + x] = v;
            if (nxt) nxt[c * nxt_plane + y * nxt_pow + x] = v;
        }
    }
}

template <class L>
__device__ __forceinline__ void chain_dispatch(int nt, const float* wa, const float* gys, const float* xact, float* gx, float* nxt,
                                               int nxt_pow, int nxt_plane, int cls, int ct, int t0, int lane) {
    if (nt == 1) chain_item<L, 1>(wa, gys, xact, gx, nxt, nxt_pow, nxt_plane, cls, ct, t0, lane);
    else if (nt == 2) chain_item<L, 2>(wa, gys, xact, gx, nxt, nxt_pow, nxt_plane, cls, ct, t0, lane);
    else if (nt == 3) chain_item<L, 3>(wa, gys, xact, gx, nxt, nxt_pow, nxt_plane, cls, ct, t0, lane);
    else if (nt == 4) chain_item<L, 4>(wa, gys, xact, gx, nxt, nxt_pow, nxt_plane, cls, ct, t0, lane);
}

__global__ void __launch_bounds__(kChainNT)
img_chain_kernel(const float* __restrict__ g5, const float* __restrict__ wa5, const float* __restrict__ wa4,
                 const float* __restrict__ wa3, const float* __restrict__ act4, const float* __restrict__ act3,
                 const float* __restrict__ act2, float* __restrict__ g4, float* __restrict__ g3, float* __restrict__ g2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t b = blockIdx.x;
    PHR_INIT(3, VAR_PH_THREAD);
    // conv 3's filter image (72 KB, the one the longest stage streams 2.4 times over) -> LDS by LDS-DMA, 1 KiB per wave
    // instruction; it lands during the first two stages
#pragma unroll
    for (int i = 0; i < (kW3Floats / 256 + 15) / 16; ++i) {
        const int k = wave + 16 * i;
        if (k < kW3Floats / 256)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wa3 + k * 256 + 4 * lane),
                                             (__attribute__((address_space(3))) void*)(lds + kW3 + k * 256), 16, 0, 0);
    }
    // zero every tile (pads stay zero; data cells are written by the stages), then stage gact5
    for (int e = tid; e < kChainTileFloats / 4; e += kChainNT) ((float4*)lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < 64 * 9) {
        const int n = tid / 9, p = tid - n * 9;
        lds[kG5 + n * L5::PLANE + (p / 3) * L5::POW + (p % 3)] = g5[b * 64 * 9 + tid];
    }
    __syncthreads();
    PHR(0);
    // ---- conv 5: 4 classes x 4 channel tiles, 9 pixels each (one tile): one item per wave ----
    chain_item<L5, 1>(wa5, lds + kG5, act4 + b * 64 * 36, g4 + b * 64 * 36, lds + kG4, L4::POW, L4::PLANE, wave >> 2, wave & 3, 0, lane);
    PHR(1);
    __syncthreads();
    PHR(2);
    // ---- conv 4: classes of 36 / 30 / 30 / 25 pixels (3 / 2 / 2 / 2 tiles) x 4 channel tiles: one item per wave ----
    {
        const int cls = wave >> 2;
        chain_dispatch<L4>(cls == 0 ? 3 : 2, wa4, lds + kG4, act3 + b * 64 * 121, g3 + b * 64 * 121, lds + kG3, L3::POW, L3::PLANE,
                           cls, wave & 3, 0, lane);
    }
    PHR(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of the filter DMA has landed
    __syncthreads();
    PHR(4);
    // ---- conv 3: classes of 121 / 110 / 110 / 100 pixels (8 / 7 / 7 / 7 tiles) x 2 channel tiles; per channel tile 8 waves:
    //      even-even 8 tiles in two passes of 4 (16 k-steps each), even-odd and odd-even 4 + 3, odd-odd 3 + 2 + 2 ----
    {
        // Item sizes in MFMAs: ee 128 (two passes of 4 tiles x 16 k-steps), eo / oe 128 + 96, oo 192 + 128 + 128.  Waves w and w + 4
        // share a SIMD (wave -> SIMD is cyclic): channel tile 0 pairs them (oo_a, eo_b) 288, (ee, oo_b) 256, (eo_a, oo_c) 256,
        // (oe_a, oe_b) 224 on SIMDs 0..3, channel tile 1 takes the same pairs on SIMDs 3, 2, 1, 0 -- 512 MFMAs on every SIMD.
        const int ct = wave >> 3, w8 = wave & 7;
        const int simd = ct ? 3 - (w8 & 3) : (w8 & 3), second = w8 >> 2;
        // item of (simd slot, second): {oo_a, ee, eo_a, oe_a} then {eo_b, oo_b, oo_c, oe_b}
        const int item = second * 4 + simd;
        const float* xa = act2 + b * 32 * 441;
        float* go = g2 + b * 32 * 441;
        const float* wa = lds + kW3;
        const float* gs = lds + kG3;
        if (item == 1) {                                        // ee: 8 tiles
            chain_item<L3, 4>(wa, gs, xa, go, nullptr, 0, 0, 0, ct, 0, lane);
            chain_item<L3, 4>(wa, gs, xa, go, nullptr, 0, 0, 0, ct, 4, lane);
        } else if (item == 2) chain_item<L3, 4>(wa, gs, xa, go, nullptr, 0, 0, 1, ct, 0, lane);      // eo tiles [0, 4)
        else if (item == 4) chain_item<L3, 3>(wa, gs, xa, go, nullptr, 0, 0, 1, ct, 4, lane);       // eo tiles [4, 7)
        else if (item == 3) chain_item<L3, 4>(wa, gs, xa, go, nullptr, 0, 0, 2, ct, 0, lane);       // oe tiles [0, 4)
        else if (item == 7) chain_item<L3, 3>(wa, gs, xa, go, nullptr, 0, 0, 2, ct, 4, lane);       // oe tiles [4, 7)
        else if (item == 0) chain_item<L3, 3>(wa, gs, xa, go, nullptr, 0, 0, 3, ct, 0, lane);       // oo tiles [0, 3)
        else if (item == 5) chain_item<L3, 2>(wa, gs, xa, go, nullptr, 0, 0, 3, ct, 3, lane);       // oo tiles [3, 5)
        else chain_item<L3, 2>(wa, gs, xa, go, nullptr, 0, 0, 3, ct, 5, lane);                      // oo tiles [5, 7)
    }
    PHR(5);
    PHR_FLUSH();
}
}  // namespace

// data gradients of conv 5, 4, 3 at 84 x 84: consumes gact[5] and act[2..4], leaves gact[4], gact[3], gact[2]
int launch_img_bwd_chain(var_ctx* c, hipStream_t s, int B) {
    ProfScope prof(c, s, TAG_IMG_DGRAD0 + 2);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    constexpr int LDS_BYTES = kChainLdsFloats * 4;
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const PackLayout& K = c->kl;
    hipLaunchKernelGGL(img_chain_kernel, dim3(B), dim3(kChainNT), LDS_BYTES, s, c->gact[5], c->wpack + K.img_a[4],
                       c->wpack + K.img_a[3], c->wpack + K.img_a[2], c->act[4], c->act[3], c->act[2], c->gact[4], c->gact[3],
                       c->gact[2]);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
