// Data gradients of conv 5 -> conv 4 -> conv 3 of the image CNN in ONE kernel, one 16-wave workgroup per image (autograd of
// models/pretext/arm_pretext_model.py:13-18 under loss.backward(), VAR/pretext_VAR.py:68): the mirror image of the fused
// forward img_mid3.hip.  From the third conv on an image's gradients are small (gact5 2 KB, gact4 9 KB, gact3 31 KB): they
// stay in LDS between the layers, so the three dependent launches of round 2 are one chain without launch boundaries.
//
// Per layer  gx = (W^T (*) gy) . (x > 0):  stride-2 transposed conv as 2 x 2 parity classes, each a dense implicit GEMM
// D[c][pixel] on v_mfma_f32_16x16x4_f32 (16 channels x 16 pixels x 4 gy channels); the gy operand comes from the layer's
// LDS tile (zero row / column for the +1 shifts of the odd classes).
//
// Second form (round 4).  The first one (34 us, 37 % MFMA-busy) issued ~1100 vector-memory wave-instructions per image -- 4-byte
// ReLU-mask loads and 4-byte result stores per accumulator register, a 16-byte filter load per four k-steps, three ahead -- and a
// CU accepts one every ~35 cycles while a filter piece takes ~1.6 K cycles to arrive: the memory pipe, not the matrix pipe, set
// its time.  This form issues ~590, all 16 bytes per lane, and asks for them far ahead:
//   * conv 5 / conv 4: wave = (parity class, 16-channel tile) reads every filter byte it needs exactly once, so a wave's filter
//     slices (4 / 8 / 8 / 16 KiB-pieces by class) are requested by the wave itself into private LDS slots (one per piece: together
//     the 144 KB of a layer), 4-8 pieces ahead -- conv 5's from the prologue on, conv 4's into the same slots as soon as the wave's
//     conv-5 loop has emptied them.  The hand-over is the wave's own counted s_waitcnt vmcnt, no barrier; a class's taps, piece
//     count and with them every count it waits for are compile-time constants (the class is a template parameter, the loops are
//     unrolled: with run-time counts the waits were a tree of scalar branches, ~900 cycles per step).  Both stages are bound by
//     what a CU takes in (measured here: 15-18 B/clk, L2-warm or not), not by their 14 K cycles of matrix work;
//   * conv 3's filter (each slice serves up to eight pixel tiles) is resident, one half per pass of 16 output channels; the first
//     half is requested by the waves of the two shortest classes, into their own slots, while the longest is still in conv 4;
//   * a stage leaves its raw results in an LDS tile; a short pass of all 1024 lanes then reads the tile in NCHW order, applies
//     the ReLU mask (the forward activation, fetched as float4 a stage earlier and kept as four bits), writes the masked values
//     back for the next stage and stores them to HBM as float4 (the weight gradients read them); the stores, the next masks and
//     the second filter half leave behind the MFMA groups of conv 3, one or two instructions at a time.
// Register loads are consumed (turned into mask bits) at points where the wave's DMA queue is empty: hipcc's own wait for a
// parked load is vmcnt(0) more often than not, and would drain a stage's filter pieces there.
// 84 x 84 (act2 21 x 21) and 96 x 96 (24 x 24): the geometry is a template parameter; conv 3's passes are dealt per size.
#include <stdlib.h>

#include <type_traits>

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
constexpr int kChainNT = 1024;

template <int H2_>
struct ChCfg {
    static constexpr int H2 = H2_, H3 = (H2 - 1) / 2 + 1, H4 = (H3 - 1) / 2 + 1, H5 = (H4 - 1) / 2 + 1;
    static constexpr int PL2 = H2 * H2, P3 = H3 * H3, P4 = H4 * H4, P5 = H5 * H5;
    // padded gy tiles [n][HO + 1][HO + 1], last row / column zero (gact5 is flat: its edge reads are masked instead)
    static constexpr int W4 = H4 + 1, T4 = W4 * W4, W3 = H3 + 1, T3 = W3 * W3;
    // LDS map (floats).  conv 5 / conv 4: the filter slots (144 pieces: 4 / 8 / 8 / 16 per wave by class), gact5 flat, gact4 tile
    static constexpr int SL = 0, NSL = 144;
    static constexpr int G5F = SL + NSL * 256, G4 = G5F + 768, END54 = G4 + 64 * T4;
    // conv 3 (units of 256 floats = one filter slot): its filter's first half (output channels 0..15: 36 pieces) is requested by the
    // waves of classes 0 and 1 as soon as THEY are through conv 4, into their own slots (0..15 four per wave, 16..47 five of a
    // wave's eight); the second half goes where classes 1 and 2 were (48..79, and the sixth slot of waves 4..7); the gact3 tile and
    // the flat result tile of a pass lie in class 3's slots (80..143)
    static constexpr int w3a(int p) { return SL + 256 * (p < 16 ? p : 16 + 8 * ((p - 16) / 5) + (p - 16) % 5); }
    static constexpr int w3b(int p) { return SL + 256 * (p < 32 ? 48 + p : 16 + 8 * (p - 32) + 5); }
    static constexpr int G3 = SL + 80 * 256, G3F = 64 * T3, G2T = G3 + G3F, END3 = G2T + 16 * PL2;
    static constexpr int LDS_FLOATS = 40960, LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int N4Q = 64 * P4 / 4, N3Q = 64 * P3 / 4, N2Q = 16 * PL2 / 4;   // float4 of gact4, gact3, half of gact2
    static constexpr int R3R = (N3Q + kChainNT - 1) / kChainNT, R2 = (N2Q + kChainNT - 1) / kChainNT;
    static constexpr int NPAD3 = 64 * (2 * W3 - 1);                // pad cells of the gact3 tile
    static_assert(H4 == 6 && H5 == 3, "the maps this kernel is laid out for");
    static_assert(N4Q <= kChainNT, "gact4: one float4 per lane");
    // (96 x 96: the result tile runs on into the gact5 / gact4 area behind the slots -- dead by then)
    static constexpr int MING = H2 == 21 ? 4 : 8;                  // MFMA groups of the shortest first item of a conv-3 pass (below)
    static_assert(END54 <= LDS_FLOATS && END3 <= LDS_FLOATS && G3 + G3F <= SL + NSL * 256 && G3F % 4 == 0, "LDS plan; 16-byte alignment");
    static_assert(H2 == 21 || H2 == 24, "the conv-3 passes are dealt for these two maps");
    static_assert((64 * P4) % 4 == 0 && (64 * P3) % 4 == 0 && (16 * PL2) % 4 == 0, "whole float4");
};

__device__ __forceinline__ void dma16(const float* g, float* l) {      // g: this lane's 16 bytes; l: the piece's (wave-uniform) LDS base
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier for LDS hand-overs that leaves the wave's DMA pieces in flight (__syncthreads() carries a fence -> vmcnt(0))
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {               // f(integral_constant<int, I>) for I in [I, N): a loop whose index is a constant expression
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ unsigned mask_bits(f32x4c m) {
    return (m[0] > 0.f ? 1u : 0u) | (m[1] > 0.f ? 2u : 0u) | (m[2] > 0.f ? 4u : 0u) | (m[3] > 0.f ? 8u : 0u);
}

// A wave's filter pieces of a layer are requested D at first, then X other vector-memory instructions, then piece j + D when piece j has
// been consumed.  The number of instructions BEHIND piece i in the queue when the step before waits for it: the D - 2 pieces after
// it that are on their way (fewer at the end of the n), and the X if piece i is one of the first D.
constexpr int behind_cnt(int i, int X, int D, int n) { return (D - 2 < n - 1 - i ? D - 2 : n - 1 - i) + (i < D ? X : 0); }

// Parity class CLS = 2 py + px of a stride-2 transposed 3 x 3 conv: its taps ((py ? 2 : 1) * (px ? 2 : 1) of the nine), and whether
// a tap reads gy row j + 1 (dy) / column i + 1 (dx)
template <int CLS>
struct Par {
    static constexpr int py = CLS >> 1, px = CLS & 1, ntap = (py ? 2 : 1) * (px ? 2 : 1), n5 = 4 * ntap;
    static constexpr int ky(int tt) { return py ? ((px ? tt >> 1 : tt) ? 2 : 0) : 1; }
    static constexpr int kx(int tt) { return px ? ((tt & 1) ? 2 : 0) : 1; }
    static constexpr int tap(int tt) { return ky(tt) * 3 + kx(tt); }
    static constexpr int dy(int tt) { return ky(tt) == 0 ? 1 : 0; }
    static constexpr int dx(int tt) { return kx(tt) == 0 ? 1 : 0; }
};

// One item of conv 3: NT pixel tiles [t0, t0 + NT) of parity class CLS for the pass's 16 channels; filter slices from the resident
// image `wf` ([tap][group][lane][4 k-steps]), gy from the gact3 tile; raw results into the flat tile `out` [16][H2][H2].
// Fully unrolled (4 ntap groups); operands of group g + 1 are fetched before the MFMAs of group g; short items (NT <= 2) keep two
// accumulators per tile (even / odd groups): one chain of dependent MFMAs runs at the instruction's latency, not at its issue rate.
// mid(g) runs behind the MFMAs of group g (every item has at least four groups): stores and loads of the caller, one or two
// instructions at a time -- a CU accepts a vector-memory instruction every ~35 cycles, and a wave that issues one behind a hundred
// others stands still until its turn: between the stages that was 4 K cycles with the matrix pipe idle.
template <class C, int NT, int CLS, int PASS, class MID>
__device__ __forceinline__ void c3_item(const float* lds, int t0, int lane, MID&& mid) {
    const float* gys = lds + C::G3;
    float* out = const_cast<float*>(lds) + C::G2T;
    using P = Par<CLS>;
    constexpr int NA = NT <= 2 ? 2 : 1, NG = P::n5;
    const int q = lane >> 4, l15 = lane & 15;
    constexpr int NI = (C::H2 + 1 - P::px) / 2, NJ = (C::H2 + 1 - P::py) / 2, npx = NI * NJ;
    int lb[NT], oa[NT];
    bool ok[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        int m = (t0 + u) * 16 + l15;
        ok[u] = m < npx;
        if (!ok[u]) m = 0;
        const int pj = m / NI, pi = m - pj * NI;
        lb[u] = q * C::T3 + pj * C::W3 + pi;
        oa[u] = 4 * q * C::PL2 + (2 * pj + P::py) * C::H2 + 2 * pi + P::px;
    }
    f32x4c acc[NA][NT];
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[k][u] = {0.f, 0.f, 0.f, 0.f};
    f32x4c a[NT <= 2 ? 2 : 1];
    float bb[NT <= 2 ? 2 : 1][4][NT];
    auto fetch = [&](int buf, int g) {                           // g: compile-time after unrolling
        const int tt = g >> 2, sg = g & 3;
        a[buf] = *(const f32x4c*)(lds + (PASS ? C::w3b(P::tap(tt) * 4 + sg) : C::w3a(P::tap(tt) * 4 + sg)) + 4 * lane);
        const int off = P::dy(tt) * C::W3 + P::dx(tt) + 16 * sg * C::T3;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NT; ++u) bb[buf][s][u] = gys[lb[u] + off + 4 * s * C::T3];
    };
    constexpr bool PIPE = NT <= 2;                                // (four tiles: 16 MFMAs per group cover the next group's reads by themselves)
    if constexpr (PIPE) fetch(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int cur = PIPE ? g & 1 : 0;
        // the waves of a SIMD are served oldest first: a wave lowers its priority as it gets on, so that the four stay together
        if (g == 0 || (g * 4) / NG != ((g - 1) * 4) / NG) {
            const int pr = 3 - (g * 4) / NG;
            if (pr == 3) __builtin_amdgcn_s_setprio(3);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (!PIPE) fetch(0, g);
        f32x4c av = a[cur];
        float vv[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                vv[s][u] = bb[cur][s][u];
                if constexpr (PIPE) asm volatile("" : "+v"(vv[s][u]));
            }
        if constexpr (PIPE) asm volatile("" : "+v"(av));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIPE)
            if (g + 1 < NG) fetch(cur ^ 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < NT; ++u)
                acc[NA == 2 ? (g & 1) : 0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], vv[s][u], acc[NA == 2 ? (g & 1) : 0][u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NT; ++u) asm volatile("" : "+v"(acc[NA == 2 ? (g & 1) : 0][u]));
        __builtin_amdgcn_sched_barrier(0);
        mid(g);                                                   // the caller's vector-memory work, a group at a time behind the MFMAs
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[oa[u] + r * C::PL2] = NA == 2 ? acc[0][u][r] + acc[NA - 1][u][r] : acc[0][u][r];
    }
    __builtin_amdgcn_s_setprio(0);
}

template <class C>
__global__ void __launch_bounds__(kChainNT)
img_chain_kernel(const float* __restrict__ g5, const float* __restrict__ wpack, int o5, int o4, int o3, const float* __restrict__ act4,
                 const float* __restrict__ act3, const float* __restrict__ act2, float* __restrict__ g4, float* __restrict__ g3,
                 float* __restrict__ g2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const size_t b = blockIdx.x;
    const int cls = wave >> 2, ct = wave & 3;                     // conv 5 / conv 4: this wave's parity class and 16-channel tile
    // its filter slots: waves 0..3 four each, 4..11 eight, 12..15 sixteen
    const int sl = C::SL + 256 * (wave < 4 ? 4 * wave : wave < 12 ? 16 + 8 * (wave - 4) : 80 + 16 * (wave - 12));
    PHR_INIT(3, VAR_PH_THREAD);
    unsigned mb3[C::R3R];
    f32x4c m2a[C::R2], m2b[C::R2];                               // ReLU masks of the two conv-3 passes (act2, 16 channels each)
    auto load_m2 = [&](f32x4c (&m2)[C::R2], int pass) {
#pragma unroll
        for (int r = 0; r < C::R2; ++r) {
            const int e = r * kChainNT + tid;
            m2[r] = ((const f32x4c*)(act2 + (b * 32 + 16 * pass) * C::PL2))[e < C::N2Q ? e : C::N2Q - 1];
        }
    };

    // Prologue, conv 5, conv 4 of one parity class (everything about a class -- its taps, the number of its filter pieces and with it
    // every vmcnt this wave waits for -- is a compile-time constant: the loops are unrolled, the waits immediates)
    auto front = [&](auto cls_c) {
        using P = Par<decltype(cls_c)::value>;
        constexpr int n5 = P::n5;
        // ---- prologue: gact5 (flat, three pieces), this wave's conv-5 filter pieces, the ReLU mask of gact4 (act4 in NCHW order:
        //      one float4 per lane; lanes past the end repeat the last one, so that every wave issues the same instructions) ----
        if (wave < 3) {
            const int f = (wave * 64 + lane) * 4;
            dma16(g5 + b * 64 * 9 + (f < 64 * 9 ? f : 64 * 9 - 4), lds + C::G5F + wave * 256);
        }
        // pieces in flight per wave and layer: D (all of a short class's, half of a long one's: more than ~80 KB on their way
        // to one CU and they arrive SLOWER); piece j + D is requested when piece j has been consumed
        constexpr int D = n5 <= 4 ? n5 : n5 / 2;
        auto issue_piece = [&](int ow, int j) {                   // slot j <- piece j of the layer at wpack + ow (j: a constant after unrolling)
            dma16(wpack + ow + ((P::tap(j >> 2) * 4 + ct) * 4 + (j & 3)) * 256 + 4 * lane, lds + sl + j * 256);
        };
        auto issue_first = [&](int ow) {
#pragma unroll
            for (int j = 0; j < D; ++j) issue_piece(ow, j);
        };
        fence();
        issue_first(o5);
        fence();
        constexpr int XM4 = 1;                                    // (one instruction: the count the waits below add for it)
        const f32x4c m4 = ((const f32x4c*)(act4 + b * 64 * C::P4))[tid < C::N4Q ? tid : C::N4Q - 1];
        fence();
        if (tid < 64 * C::T4 / 4) ((f32x4c*)(lds + C::G4))[tid] = f32x4c{0.f, 0.f, 0.f, 0.f};  // the gact4 tile: its pads stay zero
        if (wave < 3) wait_vm<D + XM4>();                         // gact5's piece; behind it: the filter pieces, the mask
        bar();
        PHR(0);

        // ---- conv 5: 4 classes x 4 channel tiles, 9 pixels each (one tile): one item per wave; two accumulators (even / odd pieces) ----
        const bool ok = l15 < 9;
        const int m = ok ? l15 : 0, pj = m / 3, pi = m - 3 * pj;
        const bool zy = pj == 2, zx = pi == 2;                   // the flat map's edge: row j + 1 / column i + 1 read as zero
        {
            const int lb = C::G5F + q * 9 + pj * 3 + pi;
            f32x4c acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            f32x4c a[2];
            float bb[2][4];
            auto fetch = [&](int buf, int j) {                   // piece j has landed
                a[buf] = *(const f32x4c*)(lds + sl + j * 256 + 4 * lane);
                const int off = P::dy(j >> 2) * 3 + P::dx(j >> 2) + 16 * (j & 3) * 9;
#pragma unroll
                for (int s = 0; s < 4; ++s) bb[buf][s] = lds[lb + off + 4 * s * 9];
            };
            wait_vm<D - 1 + XM4>();
            fetch(0, 0);
            static_for<0, n5>([&](auto jc) {
                constexpr int j = decltype(jc)::value, k = j & 1;
                const bool z = (P::dy(j >> 2) && zy) || (P::dx(j >> 2) && zx);
                f32x4c av = a[k];
                float vv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { vv[s] = z ? 0.f : bb[k][s]; asm volatile("" : "+v"(vv[s])); }
                asm volatile("" : "+v"(av));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j + 1 < n5) {
                    wait_vm<behind_cnt(j + 1, XM4, D, n5)>();
                    fetch(k ^ 1, j + 1);
                }
                if constexpr (j + D < n5) {
                    fence();
                    issue_piece(o5, j + D);                       // (slot j + D: free, nobody has used it yet)
                    fence();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], vv[s], acc[k], 0, 0, 0);
                asm volatile("" : "+v"(acc[k]));                   // (keeps the MFMAs in their step: unpinned, hipcc gathers them behind the last fetch)
                __builtin_amdgcn_sched_barrier(0);
            });
            PHR(1);
            // the mask becomes bits here, with nothing else in the queue; then conv 4's filter pieces (into the slots this loop has
            // emptied) and the ReLU mask of gact3 behind them
            unsigned mb4 = mask_bits(m4);
            asm volatile("" : "+v"(mb4)::"memory");
            issue_first(o4);
            fence();
            f32x4c m3[C::R3R];
#pragma unroll
            for (int r = 0; r < C::R3R; ++r) {
                const int e = r * kChainNT + tid;
                m3[r] = ((const f32x4c*)(act3 + b * 64 * C::P3))[e < C::N3Q ? e : C::N3Q - 1];
            }
            fence();
            if (ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lds[C::G4 + (16 * ct + 4 * q + r) * C::T4 + (2 * pj + P::py) * C::W4 + 2 * pi + P::px] = acc[0][r] + acc[1][r];
            }
            PHR(2);
            bar();
            // gact4: mask, back into the tile, out
            {
                const int e = tid < C::N4Q ? tid : C::N4Q - 1;
                f32x4c o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 4 * e + i, c = f / C::P4, rem = f - c * C::P4, y = rem / C::H4, x = rem - y * C::H4;
                    const int ad = C::G4 + c * C::T4 + y * C::W4 + x;
                    o[i] = (mb4 >> i) & 1 ? lds[ad] : 0.f;
                    lds[ad] = o[i];
                }
                fence();
                ((f32x4c*)(g4 + b * 64 * C::P4))[e] = o;
                fence();
            }
            bar();
            PHR(3);

            // ---- conv 4: one item per wave: its class's pixels (84: 36 / 30 / 30 / 25 -> 3 / 2 / 2 / 2 tiles) x its channel tile.
            //      Behind piece j in the queue: the pieces after it, gact3's mask (R3R loads), gact4's store ----
            constexpr int NI = (C::H3 + 1 - P::px) / 2, NJ = (C::H3 + 1 - P::py) / 2, npx = NI * NJ, NT = (npx + 15) / 16;
            constexpr int NA = NT <= 2 ? 2 : 1, X = C::R3R + 1;
            int lb4[NT], oa[NT];
            bool ok4[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                int mm = u * 16 + l15;
                ok4[u] = mm < npx;
                if (!ok4[u]) mm = 0;
                const int j4 = mm / NI, i4 = mm - j4 * NI;
                lb4[u] = C::G4 + q * C::T4 + j4 * C::W4 + i4;
                oa[u] = C::G3 + (16 * ct + 4 * q) * C::T3 + (2 * j4 + P::py) * C::W3 + 2 * i4 + P::px;
            }
            f32x4c ac[NA][NT];
#pragma unroll
            for (int k = 0; k < NA; ++k)
#pragma unroll
                for (int u = 0; u < NT; ++u) ac[k][u] = {0.f, 0.f, 0.f, 0.f};
            f32x4c a4[2];
            float b4[2][4][NT];
            auto fetch4 = [&](int buf, int j) {
                a4[buf] = *(const f32x4c*)(lds + sl + j * 256 + 4 * lane);
                const int off = P::dy(j >> 2) * C::W4 + P::dx(j >> 2) + 16 * (j & 3) * C::T4;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < NT; ++u) b4[buf][s][u] = lds[lb4[u] + off + 4 * s * C::T4];
            };
            wait_vm<D - 1 + X>();
            fetch4(0, 0);
            static_for<0, n5>([&](auto jc) {
                constexpr int j = decltype(jc)::value, k = j & 1;
                f32x4c av = a4[k];
                float vv[4][NT];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < NT; ++u) { vv[s][u] = b4[k][s][u]; asm volatile("" : "+v"(vv[s][u])); }
                asm volatile("" : "+v"(av));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j + 1 < n5) {
                    wait_vm<behind_cnt(j + 1, X, D, n5)>();
                    fetch4(k ^ 1, j + 1);
                }
                if constexpr (j + D < n5) {
                    fence();
                    issue_piece(o4, j + D);
                    fence();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        ac[NA == 2 ? k : 0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], vv[s][u], ac[NA == 2 ? k : 0][u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NT; ++u) asm volatile("" : "+v"(ac[NA == 2 ? k : 0][u]));
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_setprio(0);
            // the masks become bits (queue: nothing but them and a store).  The waves of classes 0 and 1, the first ones through,
            // request the first half of conv 3's filter into their own slots; everybody: the ReLU mask of conv 3's first pass
#pragma unroll
            for (int r = 0; r < C::R3R; ++r) { mb3[r] = mask_bits(m3[r]); asm volatile("" : "+v"(mb3[r])::"memory"); }
            if constexpr (P::n5 == 4 || (P::py == 0 && P::px == 1)) {
                constexpr int NP = P::n5 == 4 ? 4 : 5;
                const int p0 = P::n5 == 4 ? 4 * wave : 16 + 5 * (wave - 4);
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int pp = p0 + k;
                    dma16(wpack + o3 + (((pp >> 2) * 2 + 0) * 4 + (pp & 3)) * 256 + 4 * lane, lds + sl + k * 256);
                }
            }
            fence();
            PHR(4);
            bar();                                                // every wave has left its slots and the gact4 tile
            for (int id = tid; id < (C::H3 % 2 == 0 ? C::NPAD3 : 0); id += kChainNT) {     // (an odd map's classes never read the pads)
                const int n = id / (2 * C::W3 - 1), kk = id - n * (2 * C::W3 - 1);
                lds[C::G3 + n * C::T3 + (kk < C::W3 ? C::H3 * C::W3 + kk : (kk - C::W3) * C::W3 + C::H3)] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                if (!ok4[u]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[oa[u] + r * C::T3] = NA == 2 ? ac[0][u][r] + ac[NA - 1][u][r] : ac[0][u][r];
            }
            // A wave that has requested filter pieces above waits for them HERE, with a wait hipcc can see (vmcnt(0) expcnt(7)
            // lgkmcnt(15)): it has thousands of cycles to spare (class 3 is still in its loop), and without it hipcc -- which lays the
            // four class bodies out as a chain, each falling through to the next one's test -- protects the next body's LDS accesses
            // and registers against these "pending" pieces with a vmcnt(0) in ITS prologue, where it drains that wave's own 8 or 16.
            if constexpr (P::n5 == 4 || (P::py == 0 && P::px == 1)) __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    };
    if (cls == 0) front(std::integral_constant<int, 0>{});
    else if (cls == 1) front(std::integral_constant<int, 1>{});
    else if (cls == 2) front(std::integral_constant<int, 2>{});
    else front(std::integral_constant<int, 3>{});
    bar();
    // gact3: mask, back into the tile; the stores leave beside the first pass of conv 3
    f32x4c o3v[C::R3R];
#pragma unroll
    for (int r = 0; r < C::R3R; ++r) {
        const int e0 = r * kChainNT + tid, e = e0 < C::N3Q ? e0 : C::N3Q - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * e + i, c = f / C::P3, rem = f - c * C::P3, y = rem / C::H3, x = rem - y * C::H3;
            const int ad = C::G3 + c * C::T3 + y * C::W3 + x;
            o3v[r][i] = (mb3[r] >> i) & 1 ? lds[ad] : 0.f;
            lds[ad] = o3v[r][i];
        }
    }
    wait_vm<0>();                                                // the first half of conv 3's filter
    bar();
    PHR(5);
    // ---- conv 3, two passes of 16 output channels; a pass = 64 (tile, tap) units of 16 MFMAs each, four per wave:
    //      even-even 8 tiles x 1 tap: two waves of 4 tiles; even-odd / odd-even 7 tiles x 2 taps: three waves of 2 tiles each, and one
    //      wave for the two last tiles; odd-odd 7 tiles x 4 taps: seven waves of one tile ----
    f32x4c o2[C::R2];                                            // the first pass's results, stored beside the second
    auto conv3_pass = [&](auto pass_c, auto&& mid) {
        constexpr int PS = decltype(pass_c)::value;
        auto none = [](int) {};
        if constexpr (C::H2 == 21) {
            if (wave < 2) c3_item<C, 4, 0, PS>(lds, 4 * wave, lane, mid);
            else if (wave < 5) c3_item<C, 2, 1, PS>(lds, 2 * (wave - 2), lane, mid);
            else if (wave < 8) c3_item<C, 2, 2, PS>(lds, 2 * (wave - 5), lane, mid);
            else if (wave == 8) {
                c3_item<C, 1, 1, PS>(lds, 6, lane, mid);
                c3_item<C, 1, 2, PS>(lds, 6, lane, none);
            } else c3_item<C, 1, 3, PS>(lds, wave - 9, lane, mid);
        } else {
            // 24 x 24: four classes of 144 pixels = 9 tiles each, 81 (tile, tap) units per pass.  Nine waves take an odd-odd tile and
            // an even-even one (4 + 1 units), six waves three tiles of an even-odd / odd-even class (6 units), one wave only carries
            // its share of the loads and stores; per SIMD (wave & 3) 21 / 21 / 22 / 17 units.
            const int simd = wave & 3, slot = wave >> 2;
            const int t = simd < 2 ? (slot < 3 ? 3 * simd + slot : -1) : simd == 2 ? (slot < 2 ? 6 + slot : -1) : (slot == 0 ? 8 : -1);
            if (t >= 0) {
                c3_item<C, 1, 3, PS>(lds, t, lane, mid);
                c3_item<C, 1, 0, PS>(lds, t, lane, none);
            } else if (wave == 12 || wave == 13 || wave == 10) c3_item<C, 3, 1, PS>(lds, wave == 12 ? 0 : wave == 13 ? 3 : 6, lane, mid);
            else if (wave == 14 || wave == 7 || wave == 11) c3_item<C, 3, 2, PS>(lds, wave == 14 ? 0 : wave == 7 ? 3 : 6, lane, mid);
            else {
#pragma unroll
                for (int g = 0; g < C::MING; ++g) mid(g);
            }
        }
    };
    auto finish2 = [&](f32x4c (&m2)[C::R2], f32x4c (&o)[C::R2]) {
#pragma unroll
        for (int r = 0; r < C::R2; ++r) {
            const int e0 = r * kChainNT + tid, e = e0 < C::N2Q ? e0 : C::N2Q - 1;
            const f32x4c v = *(const f32x4c*)(lds + C::G2T + 4 * e);
            const unsigned mb = mask_bits(m2[r]);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[r][i] = (mb >> i) & 1 ? v[i] : 0.f;
        }
    };
    auto store2 = [&](int pass, int r, const f32x4c& v) {
        const int e0 = r * kChainNT + tid, e = e0 < C::N2Q ? e0 : C::N2Q - 1;
        ((f32x4c*)(g2 + (b * 32 + 16 * pass) * C::PL2))[e] = v;
    };
    // first pass; behind its first groups: its own mask and the filter's second half (36 pieces, two or three per wave), gact3's
    // stores, the second pass's mask
    conv3_pass(std::integral_constant<int, 0>{}, [&](int g) {
        if (g == 0) {
            load_m2(m2a, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = wave + 16 * i;
                if (p < 36) dma16(wpack + o3 + (((p >> 2) * 2 + 1) * 4 + (p & 3)) * 256 + 4 * lane, lds + (p < 32 ? C::w3b(0) + p * 256 : C::w3b(32) + (p - 32) * 8 * 256));
            }
        } else if (g >= 1 && g <= C::R3R) {
            const int r = g - 1, e0 = r * kChainNT + tid, e = e0 < C::N3Q ? e0 : C::N3Q - 1;
            ((f32x4c*)(g3 + b * 64 * C::P3))[e] = o3v[r];
        } else if (g == C::R3R + 1) load_m2(m2b, 1);
    });
    static_assert(C::R3R + 2 <= C::MING && C::R2 <= C::MING, "every wave's first conv-3 item has the groups to hang these on");
    PHR(6);
    wait_vm<0>();                                                // the masks; the filter's second half
    bar();
    finish2(m2a, o2);
    bar();                                                       // the result tile is free again
    PHR(7);
    conv3_pass(std::integral_constant<int, 1>{}, [&](int g) {
        if (g < C::R2) store2(0, g, o2[g]);
    });
    PHR(8);
    bar();
    {
        f32x4c o[C::R2];
        finish2(m2b, o);
#pragma unroll
        for (int r = 0; r < C::R2; ++r) store2(1, r, o[r]);
    }
    PHR(9);
    PHR_FLUSH();
}
}  // namespace

// data gradients of conv 5, 4, 3: consumes gact[5] and act[2..4], leaves gact[4], gact[3], gact[2]
template <int H2>
static int launch_chain(var_ctx* c, hipStream_t s, int B) {
    using C = ChCfg<H2>;
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_chain_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const PackLayout& K = c->kl;
    hipLaunchKernelGGL(img_chain_kernel<C>, dim3(B), dim3(kChainNT), C::LDS_BYTES, s, c->gact[5], c->wpack, K.img_a[4], K.img_a[3],
                       K.img_a[2], c->act[4], c->act[3], c->act[2], c->gact[4], c->gact[3], c->gact[2]);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int launch_img_bwd_chain(var_ctx* c, hipStream_t s, int B) {
    ProfScope prof(c, s, TAG_IMG_DGRAD0 + 2);
    return c->H == 84 ? launch_chain<21>(c, s, B) : launch_chain<24>(c, s, B);
}
