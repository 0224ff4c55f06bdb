// Middle and end of the image CNN forward in ONE kernel, third form: conv 3, conv 4, conv 5 (32 -> 64 -> 64 -> 64 channels,
// each Conv2d 3x3 stride 2 pad 1 + bias + ReLU, models/pretext/arm_pretext_model.py:13-18) and the first Linear of the image
// head (imgTriplet, arm_pretext_model.py:46-50), one 16-wave workgroup per image, activations resident in LDS.
//
// What bounded the second form (img_fwd_mid.hip, 39 us alone, 30 % of the f32 matrix peak): one image per workgroup means the
// kernel time IS the per-image latency, and a third of it went into phases the matrix pipe cannot fill -- staging act2 through
// registers (9.6 K cycles), per-wave filter streams from L2 that each k-block waited for, K-slice folds, and 590 KB of filter
// and head weights per image (conv 5 and the head are bound by that stream, not by arithmetic).  This form moves every byte
// with LDS-DMA (global_load_lds, 1 KiB per wave instruction, no registers, no address arithmetic) and never waits for more
// than it needs:
//   * act2 comes as ONE flat copy of the image's NCHW planes (rows unpadded: a tap that falls off the map reads a neighbouring
//     cell and the operand is zeroed by a per-tap lane mask instead -- no padded tile, no zero fill, no store pass);
//   * conv 3's filter (72 KB, MFMA A-fragment pieces, PackLayout::img_f[2]) lands beside it; channel groups are outermost in
//     its K order, so the first half of the products starts when the first 16 planes and the first 36 pieces are there;
//   * conv 4, conv 5 and the head's weights (144 + 144 + 288 KB) are ONE stream per wave through a private six-slot ring:
//     K is split over the waves so that a wave consumes exactly the pieces it requested itself -- the hand-over is the
//     wave's own s_waitcnt vmcnt, there is no barrier and no flag inside the streaming loops, and the stream runs on across
//     the layer boundaries (conv 5's first pieces land during conv 4's last steps);
//   * v_mfma_f32_16x16x4_f32 everywhere (121 / 36 / 9 output pixels fill 95 / 75 / 56 % of 16-pixel tiles); conv 3: a wave owns
//     (16 channels, 2-3 pixel tiles) over the whole K, no fold; conv 4 / 5: (16 channels, all pixel tiles) over a quarter of K,
//     folded through LDS with the epilogue spread over all waves.
// 84 x 84 (act2 21 x 21) and 96 x 96 (24 x 24) are the same code.
#include <stdlib.h>

#include <type_traits>

#include "var_common.h"

namespace {
PH_DECL();
}
#define VAR_HEAD2_DEVICE_ONLY
#include "img_head2.hip"             // Head2Cfg, img_head2_body: conv 1 + conv 2, for the fused forward at the end of this file
#undef VAR_HEAD2_DEVICE_ONLY
namespace {
#ifdef VAR_PHASES
__device__ unsigned long long g_span[1024][2];          // per workgroup: s_memrealtime (100 MHz, chip-wide) at its first and last instruction
#endif
}
#ifdef VAR_PHASES
extern "C" int var_debug_spans_mid3(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 2 * n) == hipSuccess ? 0 : -1;
}
extern "C" int var_debug_phases_mid3(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
namespace {
typedef float f32x4q __attribute__((ext_vector_type(4)));
// the embedding's finish inside the kernel (heads.hip: heads_finish_kernel's arithmetic on this image's row): b1 = nullptr -> none
struct MidFinish { const float* b1; float* emb_raw; float* emb; float* out; };
constexpr int M3_NT = 1024;                    // 16 waves

template <int H2_>
struct M3Cfg {
    static constexpr int H2 = H2_, H3 = (H2 - 1) / 2 + 1, H4 = (H3 - 1) / 2 + 1, H5 = (H4 - 1) / 2 + 1;
    static constexpr int PL2 = H2 * H2, P3 = H3 * H3, P4 = H4 * H4, P5 = H5 * H5;
    static constexpr int X2F = 32 * PL2;                           // floats of an image's act2
    static constexpr int NX = (X2F * 4 + 1023) / 1024;             // its 1-KiB pieces (the last one may be partial)
    static constexpr int XH = (16 * PL2 * 4 + 1023) / 1024;        // pieces that cover channels 0..15
    static constexpr int NPT3 = (P3 + 15) / 16;                    // conv 3 pixel tiles: 8 / 9
    static constexpr int T3 = NPT3 % 4 == 0 ? NPT3 / 4 : 3;        // per wave: 2 (16 waves) / 3 (12 waves)
    static constexpr int NW3 = (NPT3 / T3) * 4;
    static constexpr int T4 = (P4 + 15) / 16;                      // conv 4 pixel tiles: 3
    // LDS map (floats).  Region A: the activations; region RING: 16 private rings of D pieces; CONST: biases, head layer 2.
    // During conv 3 act2 (from X2O) and conv 3's filter (W3O) lie across A and the rings.
    static constexpr int ACT3 = 64, ACT4 = ACT3 + 9216, ACT5 = ACT4 + 2304, A_END = ACT5 + 576;
    static constexpr int D = 6, RING = 12288, RING_W = D * 256;
    static constexpr int CONSTS = 39936, LDS_FLOATS = 40960, LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int C_B3 = CONSTS, C_B4 = C_B3 + 64, C_B5 = C_B4 + 64, C_HB0 = C_B5 + 64, C_HW1 = C_HB0 + 128;
    // conv 3's filter in two halves: pieces 0..35 (channels 0..15) high up -- above everything img_head2_kernel uses, so that the
    // fused forward (img_fwd_all_kernel) can request them while conv 1 / conv 2 still run -- pieces 36..71 behind act2
    static constexpr int X2O = 128, W3B = X2O + NX * 256, W3A = 27904;
    static constexpr int NE = NX + 72, PPW = NE / 16;              // prologue pieces, per wave
    static constexpr int NE1 = XH + 36;                            // entries the first half of conv 3 needs
    static constexpr int F1 = (NE1 + 15) / 16;                     // ... per wave, at most
    // fold scratch of conv 4 (9 parked registers per lane and wave), conv 5 and the head: the act3 area, dead by then
    static constexpr int RED = ACT3;
    static_assert(H4 == 6 && H5 == 3 && P5 == 9, "the maps this kernel is laid out for");
    static_assert(NE % 16 == 0 && NPT3 % T3 == 0, "whole prologue rounds, whole tiles per wave");
    static_assert(64 * P3 <= 9216 && A_END <= RING && RING + 16 * RING_W <= CONSTS, "LDS plan");
    static_assert(W3B + 9216 <= W3A && W3A + 9216 <= CONSTS && C_HW1 + 384 <= LDS_FLOATS, "conv 3 phase fits in front of the constants");
    static_assert(16 * 9 * 64 <= 9216 && X2O >= H2 + 2, "fold scratch; guard in front of act2 for the row -1 reads");
};

__device__ __forceinline__ void dma16(const float* g, float* l) {      // g: this lane's 16 bytes; l: the piece's (wave-uniform) LDS base
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void set_prio(int p) {                       // (p is a constant wherever this is called from unrolled code)
    if (p <= 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Workgroup barrier for LDS hand-overs that leaves the wave's LDS-DMA pieces in flight: __syncthreads() carries a fence that hipcc
// lowers to s_waitcnt vmcnt(0), which would drain the weight stream at every layer boundary.  Everything the waves hand each
// other in this kernel goes through LDS (lgkmcnt); what arrives by DMA is waited for explicitly (wait_vm) by the wave that asked.
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The constants block (biases, the head's second layer: four pieces) and the first half of conv 3's filter (36 pieces), requested
// by EIGHT waves (wb = 0..7, five pieces each): the fused forward runs this from the conv-2 waves of img_head2_body before their
// first barrier -- both land above everything that body keeps in LDS.
template <class C>
__device__ __forceinline__ void mid3_early(const float* __restrict__ wa3, const float* __restrict__ params, int o_b3, int o_b4, int o_b5,
                                           int o_hb0, int o_hw1, int wb, int lane) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int e = wb + 8 * r;                                  // e < 4: constants piece e; else filter piece e - 4
        const bool isc = e < 4;
        const int fo = 4 * ((e & 3) * 64 + lane);
        // (the five arrays live in ONE parameter arena: an integer select, no pointer select)
        const int offc = fo < 64 ? o_b3 + fo : fo < 128 ? o_b4 + (fo - 64) : fo < 192 ? o_b5 + (fo - 128)
                       : fo < 320 ? o_hb0 + (fo - 192) : fo < 704 ? o_hw1 + (fo - 320) : o_hw1;
        const long off = isc ? (long)offc : (long)(e - 4) * 256 + 4 * lane;
        dma16((isc ? params : wa3) + off, lds + (isc ? C::CONSTS + e * 256 : C::W3A + (e - 4) * 256));
    }
}

// PRE: mid3_early has run and landed (fused forward); b: the image of this workgroup
template <class C, bool PRE>
__device__ __forceinline__ void img_mid3_body(const float* __restrict__ x2, const float* __restrict__ wa3, const float* __restrict__ wa4,
                                              const float* __restrict__ wa5, const float* __restrict__ params, int o_b3, int o_b4, int o_b5,
                                              int o_hb0, int o_hw1, float* __restrict__ y3, float* __restrict__ y4, float* __restrict__ y5,
                                              const float* __restrict__ hw0t, float* __restrict__ hid, float* __restrict__ part, const size_t b, unsigned* sig = nullptr,
                                              MidFinish fin = MidFinish{nullptr, nullptr, nullptr, nullptr}) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, i15 = lane & 15;
    const int nt = wave & 3, kq = wave >> 2;                     // 16-channel tile; K quarter (conv 4, 5) / pixel group (conv 3)
    PHR_INIT(5, VAR_PH_THREAD);
#ifdef VAR_PHASES
    if (tid == 0) g_span[blockIdx.x][0] = wall_clock64();
#endif
    // prologue pieces per wave, and how many of them the first half of conv 3 waits for
    constexpr int PPW = PRE ? (C::NX + 36 + 15) / 16 : C::PPW, F1 = PRE ? (C::XH + 15) / 16 : C::F1;
    if constexpr (!PRE) {
        // ---- constants (biases, the head's second layer) by LDS-DMA as well: a value parked in a register until the first barrier
        //      gave hipcc a pending load to wait for at an unrelated instruction (vmcnt(0): the whole prologue).  The block is four
        //      pieces; every wave requests piece wave & 3 (identical bytes from four waves), first, so that it is the oldest in its queue
        const int fo = 4 * ((wave & 3) * 64 + lane);
        // (the five arrays live in ONE parameter arena: an integer select, no pointer select -- as pointers hipcc built a tree of
        //  divergent branches with a kernel-argument load and a wait in every leaf)
        const int off = fo < 64 ? o_b3 + fo : fo < 128 ? o_b4 + (fo - 64) : fo < 192 ? o_b5 + (fo - 128)
                      : fo < 320 ? o_hb0 + (fo - 192) : fo < 704 ? o_hw1 + (fo - 320) : o_hw1;
        dma16(params + off, lds + C::CONSTS + (wave & 3) * 256);
    }
    {
        // (branch-free: behind a branch hipcc loses count of the pieces in flight and waits for all of them at the next use.
        //  The last piece of a 21 x 21 map reads 896 bytes past the image: the next image's act2 or, for the last image, the
        //  workspace block that follows act[2] -- valid memory either way, and it lands in front of the filter, unused)
        const float* xi = x2 + b * C::X2F;
        const unsigned l4 = 4u * (unsigned)lane;
        // entry e = wave + 16 r of the list [act2 pieces of channels 0..15 | filter pieces 0..35 | rest of act2 | filter pieces 36..71]
        // (PRE: [act2 0..15 | rest of act2 | filter pieces 36..71 | the last pieces once more to fill the round]):
        // a round lies inside one segment (everything but `wave` is a compile-time constant) or across one boundary (one scalar select)
        auto seg_of = [](int e) { return PRE ? (e < C::NX ? 0 : 3) : e < C::XH ? 0 : e < C::NE1 ? 1 : e < C::NX + 36 ? 2 : 3; };
        auto piece_of = [](int sg, int e) { return PRE ? (sg == 0 ? e : e - C::NX + 36) : sg == 0 ? e : sg == 1 ? e - C::XH : sg == 2 ? e - 36 : e - C::NX; };
#pragma unroll
        for (int r = 0; r < PPW; ++r) {
            const int lo = 16 * r, sa = seg_of(lo), sb = seg_of(lo + 15);
            const int bnd = PRE ? C::NX : sa == 0 ? C::XH : sa == 1 ? C::NE1 : C::NX + 36;   // first entry of the next segment
            const bool first = sa == sb || wave < bnd - lo;
            const int sg = first ? sa : sb;
            int pc = piece_of(sg, lo) + wave;                                       // (piece_of is linear in e)
            const bool isx = (sg & 1) == 0;
            if (!isx && pc > 71) pc = 71;                                           // (PRE: the round's filler entries)
            dma16((isx ? xi : wa3) + pc * 256 + l4, lds + (isx ? C::X2O + pc * 256 : pc < 36 ? C::W3A + pc * 256 : C::W3B + (pc - 36) * 256));
        }
    }

    // ---- conv 3: wave = (channel tile nt, pixel group kq), T3 pixel tiles, whole K -----------------------------------------
    constexpr int T3 = C::T3;
    f32x4q acc3[T3];
    int base3[T3];
    bool top3[T3], left3[T3], bot3[T3], right3[T3], ok3[T3];
    int p3[T3];
#pragma unroll
    for (int t = 0; t < T3; ++t) {
        acc3[t] = {0.f, 0.f, 0.f, 0.f};
        int p = (kq * T3 + t) * 16 + i15;
        ok3[t] = p < C::P3;
        p = ok3[t] ? p : C::P3 - 1;
        p3[t] = p;
        const int oy = p / C::H3, ox = p - oy * C::H3;
        base3[t] = C::X2O + (2 * oy - 1) * C::H2 + 2 * ox - 1 + q * C::PL2;
        top3[t] = oy == 0; left3[t] = ox == 0;
        bot3[t] = (C::H2 & 1) && oy == C::H3 - 1; right3[t] = (C::H2 & 1) && ox == C::H3 - 1;
    }
    const bool act3w = wave < C::NW3;
    auto conv3_groups = [&](auto g0c, auto g1c) {
        constexpr int G0 = decltype(g0c)::value, G1 = decltype(g1c)::value;
        f32x4q a[2];
        float bb[2][4][T3];
        auto fetch = [&](int buf, int G) {
            const int cg = G / 9, tap = G % 9, toff = (tap / 3) * C::H2 + tap % 3;
            a[buf] = *(const f32x4q*)(lds + (G < 9 ? C::W3A + (G * 4 + nt) * 256 : C::W3B + ((G - 9) * 4 + nt) * 256) + 4 * lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < T3; ++t) bb[buf][j][t] = lds[base3[t] + toff + (16 * cg + 4 * j) * C::PL2];
        };
        fetch(0, G0);
#pragma unroll
        for (int G = G0; G < G1; ++G) {
            const int cur = (G - G0) & 1;
            const int tap = G % 9, ky = tap / 3, kx = tap % 3;
            // Order inside a group: (1) this group's operands out of their landing registers -- the edge selects; the wait for the
            // LDS reads sits HERE, with nothing younger in flight, (2) the next group's reads, (3) the MFMAs back to back.  With the
            // reads issued before (1) hipcc waited for them as well (lgkmcnt(0) right behind their issue, every other group); a
            // select directly in front of the MFMA that reads it costs two wait states each time (11 % of the phase).
            float vv[4][T3];
            f32x4q av = a[cur];
#pragma unroll
            for (int t = 0; t < T3; ++t) {
                const bool z = (ky == 0 && top3[t]) || (ky == 2 && bot3[t]) || (kx == 0 && left3[t]) || (kx == 2 && right3[t]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vv[j][t] = z ? 0.f : bb[cur][j][t]; asm volatile("" : "+v"(vv[j][t])); }
            }
            asm volatile("" : "+v"(av));
            __builtin_amdgcn_sched_barrier(0);
            if (G + 1 < G1) fetch(cur ^ 1, G + 1);
            // Instruction issue goes by priority, then by AGE: with equal priorities the oldest wave of a SIMD runs ahead, the
            // youngest is starved and then finishes its share alone, with nobody to cover its LDS latencies (measured: the
            // oldest wave left a 9 K-cycle phase after 9.0 K cycles, the youngest after 12.4 K).  A wave lowers its own priority
            // as it gets on: whoever is behind is served first, the four waves of a SIMD arrive together.
            if ((G * 4) / 18 != ((G - 1) * 4) / 18 || G == G0) set_prio(3 - (G * 4) / 18);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < T3; ++t) acc3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], vv[j][t], acc3[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // index math of conv 4 and conv 5 (per-lane operand bases, edge flags as bits), done here, while the prologue's pieces fly
    constexpr int T4 = C::T4;
    int base4[T4], e4[T4], base5, e5;
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        int p = t * 16 + i15;
        p = p < C::P4 ? p : C::P4 - 1;
        const int oy = p / C::H4, ox = p - oy * C::H4;
        base4[t] = C::ACT3 + (2 * oy - 1) * C::H3 + 2 * ox - 1 + (16 * kq + q) * C::P3;
        e4[t] = (oy == 0 ? 1 : 0) | (ox == 0 ? 2 : 0) | (((C::H3 & 1) && oy == C::H4 - 1) ? 4 : 0) | (((C::H3 & 1) && ox == C::H4 - 1) ? 8 : 0);
    }
    {
        const int p = i15 < C::P5 ? i15 : C::P5 - 1;             // conv 5: one pixel tile (9 of 16 lanes), same K split
        const int oy = p / C::H5, ox = p - oy * C::H5;
        base5 = C::ACT4 + (2 * oy - 1) * C::H4 + 2 * ox - 1 + (16 * kq + q) * C::P4;
        e5 = (oy == 0 ? 1 : 0) | (ox == 0 ? 2 : 0);
    }
    // (opaque: left alone hipcc sinks this arithmetic back to its first use -- the conv 3 epilogue, where every cycle counts)
#pragma unroll
    for (int t = 0; t < T4; ++t) { asm volatile("" : "+v"(base4[t])); asm volatile("" : "+v"(e4[t])); }
    asm volatile("" : "+v"(base5));
    asm volatile("" : "+v"(e5));
    {   // the output pointers out of the kernel-argument segment now (as scalar loads in the epilogues they were waited for there)
        asm volatile("" ::"s"(y3), "s"(y4), "s"(y5), "s"(hid), "s"(part), "s"(wa4), "s"(wa5), "s"(hw0t));
    }
    PHR(0);
    wait_vm<PPW - F1>();                                         // this wave's pieces of the first half have landed
    bar();
    PHR(1);
    if (act3w) conv3_groups(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
    PHR(2);
    wait_vm<0>();
    bar();
    PHR(3);
    // conv 5's filter share (9 pieces: K quarter kq of channel tile nt) goes to REGISTERS, requested here: the memory pipe has
    // just run empty, the second half of conv 3 is matrix-bound, and being the oldest requests in the queue they never figure
    // in the stream's counted waits below.  conv 5 then needs no stream of its own.
    f32x4q w5[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) w5[s] = *(const f32x4q*)(wa5 + ((s * 4 + kq) * 4 + nt) * 256 + 4 * lane);
    if (act3w) conv3_groups(std::integral_constant<int, 9>{}, std::integral_constant<int, 18>{});
    PHR(4);
    bar();                                             // act2 and conv 3's filter are dead: the rings and act3 may be written
    PHR(5);

    // ---- the weight stream of this wave: 9 pieces of conv 4, 9 of conv 5 (K quarter kq of channel tile nt), 18 of the head --
    // Memory operations retire through ONE in-order counter, and the stream's waits below count only the pieces issued after
    // the awaited one: so no global store may be issued between two pieces that are in flight together.  conv 3's stores go out
    // BEFORE the stream starts; those of conv 4, conv 5 and the head wait in registers for the end of the kernel.
    float* ring = lds + C::RING + wave * C::RING_W;
    const float* hsrc = hw0t ? hw0t : wa5;                       // (no head: the stream still runs its 27 pieces -- the waits count them -- from valid memory)
    auto issue = [&](int I) {                                    // (I is a compile-time constant at every call)
        const float* src = I < 9 ? wa4 + ((I * 4 + kq) * 4 + nt) * 256 : hsrc + (wave + 16 * (I - 9)) * 256;
        dma16(src + 4 * lane, ring + (I % C::D) * 256);
    };
    // conv 3 epilogue: bias + ReLU -> act3 in LDS (conv 4's operand; lanes beyond the map computed pixel P3 - 1 once more and
    // repeat its store).  The HBM copy the backward reads is NOT stored from here: 4-byte stores leave a CU at ~7 bytes per cycle
    // (31 KB: 4.4 K cycles with the matrix pipe idle, measured) -- the LDS tile IS the NCHW block, so it goes out as 16-byte
    // stores of whole KiB during conv 4, which does not use the memory pipe (copy_out below); act4 and act5 likewise
    if (act3w) {
        const f32x4q bias = *(const f32x4q*)(lds + C::C_B3 + 16 * nt + 4 * q);      // (one read: a read per value was a chain of eight LDS latencies)
#pragma unroll
        for (int t = 0; t < T3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = __builtin_amdgcn_fmed3f(acc3[t][r] + bias[r], 0.f, __builtin_inff());
                lds[C::ACT3 + (16 * nt + 4 * q + r) * C::P3 + p3[t]] = v;
            }
    }
    // The stream starts gently: a CU takes in ~30 bytes per cycle through its vector-memory path, and a wave that asks for more
    // than the path has room for stalls in ISSUE (measured: six pieces and conv 5's nine register loads per wave at once -- 240 KB
    // per CU -- held every wave for 6 K cycles in front of conv 4).  Three pieces now, then one piece and one register load per step.
    __builtin_amdgcn_sched_barrier(0);
    issue(0); issue(1); issue(2);
    __builtin_amdgcn_sched_barrier(0);
    PHR(6);

    // ---- conv 4: wave = (channel tile nt, K quarter kq = 16 input channels), all T4 pixel tiles ----------------------------
    f32x4q acc4[T4];
    bool top4[T4], left4[T4], bot4[T4], right4[T4];
#pragma unroll
    for (int t = 0; t < T4; ++t) {
        acc4[t] = {0.f, 0.f, 0.f, 0.f};
        top4[t] = e4[t] & 1; left4[t] = e4[t] & 2; bot4[t] = e4[t] & 4; right4[t] = e4[t] & 8;
    }
    const bool top5 = e5 & 1, left5 = e5 & 2;
    wait_vm<2>();                                                // piece 0 (younger: pieces 1, 2)
    f32x4q a_cur = *(const f32x4q*)(ring + 4 * lane), a_nxt = a_cur;
    bar();                                                       // act3 complete
    // [LDS tile of NF floats] -> HBM, one KiB per wave instruction, wave w takes pieces w, w + 16, ...
    auto copy_out = [&](const float* tile, float* dst, auto nfc) {
        constexpr int NF = decltype(nfc)::value, NP = (NF + 255) / 256;
#pragma unroll
        for (int i = 0; i < (NP + 15) / 16; ++i) {
            const int e = (wave + 16 * i) * 256 + 4 * lane;
            if (e < NF) *(f32x4q*)(dst + e) = *(const f32x4q*)(tile + e);
        }
    };
    copy_out(lds + C::ACT3, y3 + b * 64 * C::P3, std::integral_constant<int, 64 * C::P3>{});
    PHR(7);
    {
        float bb[2][4][T4];
        auto fetch4 = [&](int buf, int s) {
            const int toff = (s / 3) * C::H3 + s % 3;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < T4; ++t) bb[buf][j][t] = lds[base4[t] + toff + 4 * j * C::P3];
        };
        fetch4(0, 0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int ky = s / 3, kx = s % 3;
            // (1) this step's operands (edge selects; the LDS wait sits here), (2) piece s + 1 -- conv 5's first one after the last
            // tap -- and the next step's reads, (3) the MFMAs, the refill of the slot piece s has left behind the first of them
            float vv[4][T4];
            f32x4q av = a_cur;
#pragma unroll
            for (int t = 0; t < T4; ++t) {
                const bool z = (ky == 0 && top4[t]) || (ky == 2 && bot4[t]) || (kx == 0 && left4[t]) || (kx == 2 && right4[t]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vv[j][t] = z ? 0.f : bb[s & 1][j][t]; asm volatile("" : "+v"(vv[j][t])); }
            }
            asm volatile("" : "+v"(av));
            __builtin_amdgcn_sched_barrier(0);
            // piece s + 1; certainly younger in the queue: s = 0: piece 2 (the act3 copy-out stores are not counted: a wave has one
            // or two)
            // (pieces 3..5 leave in step 0, piece s + 5 in step s >= 1: from step 1 on three pieces are younger than the awaited one)
            if (s == 0) wait_vm<1>(); else wait_vm<3>();
            a_nxt = *(const f32x4q*)(ring + ((s + 1) % C::D) * 256 + 4 * lane);
            if (s + 1 < 9) fetch4((s + 1) & 1, s + 1);
            if (s == 0 || (s * 4) / 9 != ((s - 1) * 4) / 9) set_prio(3 - (s * 4) / 9);      // (see conv 3)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < T4; ++t) acc4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], vv[j][t], acc4[t], 0, 0, 0);
                if (j == 0) {                                    // (the MFMAs read `av`: piece s's slot may be overwritten)
                    __builtin_amdgcn_sched_barrier(0);
                    if (s == 0) { issue(3); issue(4); issue(5); } else issue(s + 5);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
        }
    }
    __builtin_amdgcn_s_setprio(0);
    PHR(8);
    float ohid = 0.f;
    // Fold the four K quarters: quarter kq owns accumulator register r = kq of every tile (channel 16 nt + 4 q + kq); a wave
    // parks the registers it does not own, sums its own over the other quarters in fixed order and runs their epilogue.
    // Branch-free (kq is wave-uniform: as branches this was 400 lines of scalar control flow): the register a wave owns is
    // written to a dump cell, and the owner's read of "itself" comes from there too and is replaced by its register.
    {
        float* red = lds + C::RED;
        float* dump = lds + C::CONSTS + 704;                     // 64 unused floats of the constants block
        bar();                                                   // every wave is done reading act3
#pragma unroll
        for (int t = 0; t < T4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sp = kq - (kq > r ? 1 : 0);            // this wave's index among owner r's three sources
                float* dst = r == kq ? dump : red + (((r * 4 + nt) * 3 + sp) * T4 + t) * 64;
                dst[lane] = acc4[t][r];
            }
        bar();
        const int n = 16 * nt + 4 * q + kq;
        const float bias = lds[C::C_B4 + n];
#pragma unroll
        for (int t = 0; t < T4; ++t) {
            const float own = kq == 0 ? acc4[t][0] : kq == 1 ? acc4[t][1] : kq == 2 ? acc4[t][2] : acc4[t][3];
            float s0 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                        // fixed order over the quarters 0..3
                const int sp = k - (k > kq ? 1 : 0);
                const float* src = k == kq ? dump : red + (((kq * 4 + nt) * 3 + sp) * T4 + t) * 64;
                const float v = src[lane];
                s0 += k == kq ? own : v;
            }
            const float o = __builtin_amdgcn_fmed3f(s0 + bias, 0.f, __builtin_inff());
            const int p = t * 16 + i15;
            if (p < C::P4) lds[C::ACT4 + n * C::P4 + p] = o;
        }
    }
    PHR(9);
    bar();                                                       // act4 complete (and the fold scratch read)
    copy_out(lds + C::ACT4, y4 + b * 64 * C::P4, std::integral_constant<int, 64 * C::P4>{});
    PHR(10);

    // ---- conv 5: wave = (channel tile nt, K quarter kq), one pixel tile ---------------------------------------------------
    // Two accumulators (even / odd taps): one chain of 36 dependent MFMAs runs at the instruction's 40-cycle latency, and the
    // waves of a SIMD are served oldest first -- four chains one after the other.  While the matrix pipe works, the head's
    // weight stream keeps flowing: each step moves one landed piece from the ring to registers (conv 5's own pieces free them)
    // and refills the slot, so that the head finds half of its 288 KB on the chip already.
    f32x4q acc5 = {0.f, 0.f, 0.f, 0.f}, acc5b = {0.f, 0.f, 0.f, 0.f};
    f32x4q hq[9];
    {
        float bb[2][4];
        auto fetch5 = [&](int buf, int s) {
            const int toff = (s / 3) * C::H4 + s % 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[buf][j] = lds[base5 + toff + 4 * j * C::P4];
        };
        fetch5(0, 0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int ky = s / 3, kx = s % 3;
            const bool z = (ky == 0 && top5) || (kx == 0 && left5);
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { vv[j] = z ? 0.f : bb[s & 1][j]; asm volatile("" : "+v"(vv[j])); }
            hq[s] = a_cur;                                        // stream piece 9 + s = piece s of the head
            asm volatile("" : "+v"(hq[s]));
            __builtin_amdgcn_sched_barrier(0);
            // piece 10 + s; younger: s = 0: pieces 11..13 (requested up to there by conv 4); s >= 1: the four after it
            if (s == 0) wait_vm<3>(); else wait_vm<4>();
            a_cur = *(const f32x4q*)(ring + ((10 + s) % C::D) * 256 + 4 * lane);
            if (s + 1 < 9) fetch5((s + 1) & 1, s + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) issue(14);                                // (fills the ring: pieces 10..15)
            issue(15 + s);                                        // into the slot piece 9 + s has left
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (s & 1) acc5b = __builtin_amdgcn_mfma_f32_16x16x4f32(w5[s][j], vv[j], acc5b, 0, 0, 0);
                else acc5 = __builtin_amdgcn_mfma_f32_16x16x4f32(w5[s][j], vv[j], acc5, 0, 0, 0);
            }
        }
        acc5 += acc5b;
    }
    PHR(11);
    {
        float* red = lds + C::RED;                               // (act3's area: conv 4's fold scratch was read before the last barrier)
        if (kq > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((kq - 1) * 4 + nt) * 4 + r) * 64 + lane] = acc5[r];
        }
        bar();
        if (kq == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc5[r];
#pragma unroll
                for (int k = 0; k < 3; ++k) v += red[((k * 4 + nt) * 4 + r) * 64 + lane];
                const int n = 16 * nt + 4 * q + r;
                v = __builtin_amdgcn_fmed3f(v + lds[C::C_B5 + n], 0.f, __builtin_inff());
                if (i15 < C::P5) lds[C::ACT5 + n * C::P5 + i15] = v;    // NCHW flatten order = the head's input
            }
        }
    }
    PHR(12);
    bar();                                                       // act5 complete
    copy_out(lds + C::ACT5, y5 + b * 64 * C::P5, std::integral_constant<int, 64 * C::P5>{});
    PHR(13);

    // ---- image head: hidden = relu(W0 a5 + b0).  Piece = two rows k, k + 1 of W0^T [k][128]: lanes 0..31 hold row k's 128
    //      columns (4 each), lanes 32..63 row k + 1's; a lane accumulates its 4 columns over the rows of its parity --------------
    if (hw0t) {
        f32x4q hacc = {0.f, 0.f, 0.f, 0.f};
        const int hl = lane >> 5;
#pragma unroll
        for (int s = 0; s < 9; ++s) hacc += hq[s] * lds[C::ACT5 + 2 * (wave + 16 * s) + hl];      // the pieces conv 5 collected
#pragma unroll
        for (int s = 9; s < 18; ++s) {
            constexpr int LAST = 26;
            const int I = 9 + s;                                  // a_cur = piece I; pieces up to 23 were requested during conv 5
            if (I + 1 <= LAST) {
                if (I + 5 <= LAST) wait_vm<4>();
                else if (LAST - I - 1 == 3) wait_vm<3>();
                else if (LAST - I - 1 == 2) wait_vm<2>();
                else if (LAST - I - 1 == 1) wait_vm<1>();
                else wait_vm<0>();
                a_nxt = *(const f32x4q*)(ring + ((I + 1) % C::D) * 256 + 4 * lane);
            }
            const float xk = lds[C::ACT5 + 2 * (wave + 16 * s) + hl];
            hacc += a_cur * xk;
            __builtin_amdgcn_sched_barrier(0);
            if (I + C::D <= LAST) issue(I + C::D);
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
        }
        // rows of both parities, then the 16 waves in fixed order
        float* hp = lds + C::RED;                                // [wave][128]
#pragma unroll
        for (int r = 0; r < 4; ++r) hacc[r] += __shfl_down(hacc[r], 32, 64);
        if (lane < 32) *(f32x4q*)(hp + wave * kHid + 4 * lane) = hacc;
        bar();
        float* hh = hp + 16 * kHid;
        if (tid < kHid) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) v += hp[w * kHid + tid];
            v = __builtin_amdgcn_fmed3f(v + lds[C::C_HB0 + tid], 0.f, __builtin_inff());
            hh[tid] = v;
            ohid = v;
        }
        bar();
        // this image's partial of the 128 -> 3 layer in the (row, 4, 4) layout the finish / rows kernels read (block 0 carries
        // the whole dot product, blocks 1..3 are zero)
        if (tid < 192) {
            const int d = tid >> 6, l = tid & 63;
            float sum = hh[l] * lds[C::C_HW1 + d * kHid + l] + hh[64 + l] * lds[C::C_HW1 + d * kHid + 64 + l];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
            if (l == 0) {
                if (sig) join_store(part + b * 16 + d, sum);
                else part[b * 16 + d] = sum;
                if (fin.b1) hh[kHid + d] = sum;                  // (hh: 128 hidden values; three more floats behind them)
            }
        }
        if (tid >= 192 && tid < 192 + 13) {
            if (sig) join_store(part + b * 16 + (tid - 192 + 3), 0.f);
            else part[b * 16 + (tid - 192 + 3)] = 0.f;
        }
        if (tid < kHid) hid[b * kHid + tid] = ohid;
        if (fin.b1) {
            // An inference / operator-API forward wants the normalised embedding: this image's row of heads_finish_kernel, here
            // (blocks 1..3 of the partial are zero: ((s + 0) + (0 + 0)) + b1 = s + b1 bit for bit) -- one launch fewer behind the kernel
            bar();
            if (tid == 0) {
                const float z = 0.f + 0.f;
                const float a = ((hh[kHid] + 0.f) + z) + fin.b1[0], bb = ((hh[kHid + 1] + 0.f) + z) + fin.b1[1], cc = ((hh[kHid + 2] + 0.f) + z) + fin.b1[2];
                const float nrm = sqrtf(a * a + bb * bb + cc * cc);
                const float den = nrm > 1e-12f ? nrm : 1e-12f;
                fin.emb_raw[b * 3 + 0] = a; fin.emb_raw[b * 3 + 1] = bb; fin.emb_raw[b * 3 + 2] = cc;
                fin.emb[b * 3 + 0] = a / den; fin.emb[b * 3 + 1] = bb / den; fin.emb[b * 3 + 2] = cc / den;
                if (fin.out) { fin.out[b * 3 + 0] = a / den; fin.out[b * 3 + 1] = bb / den; fin.out[b * 3 + 2] = cc / den; }
            }
        }
        // training step: the sound rows of the heads' backward, on the other stream, need these partials, and the image rows behind
        // this kernel the sound branch's: handed over on the device (var_common.h: join_*; heads.hip)
        if (sig) join_signal(sig, gridDim.x, sig - 4);           // (sig = jsig + 4: the sound side's words lie in front)
    } else {
        wait_vm<0>();                                            // (nothing of the stream may still be landing when the workgroup ends)
    }
    PHR(14);
    PHR_FLUSH();
#ifdef VAR_PHASES
    if (tid == 0) g_span[blockIdx.x][1] = wall_clock64();
#endif
}
template <class C>
__global__ void __launch_bounds__(M3_NT)
img_mid3_kernel(const float* __restrict__ x2, const float* __restrict__ wa3, const float* __restrict__ wa4,
                const float* __restrict__ wa5, const float* __restrict__ params, int o_b3, int o_b4, int o_b5, int o_hb0, int o_hw1,
                float* __restrict__ y3, float* __restrict__ y4, float* __restrict__ y5,
                const float* __restrict__ hw0t, float* __restrict__ hid, float* __restrict__ part, unsigned* sig, MidFinish fin) {
    img_mid3_body<C, false>(x2, wa3, wa4, wa5, params, o_b3, o_b4, o_b5, o_hb0, o_hw1, y3, y4, y5, hw0t, hid, part, blockIdx.x, sig, fin);
}

// The WHOLE image forward of one image in one workgroup: conv 1 + conv 2 (img_head2_body: role-split waves over the image's seven
// bands), then conv 3-5 + image head (the body above) -- at a full batch both kernels were one-image-per-workgroup grids anyway.
// What the fusion removes: one kernel boundary (4.5-5 us of dispatch and end-of-kernel work outside every workgroup,
// profiles/r04_inkernel_clock.txt) and most of the second kernel's cold start -- the constants and the first half of conv 3's
// filter are requested by the conv-2 waves before their first band and wait in LDS above everything conv 1 / 2 use; act2 still
// travels through memory (this CU wrote it and drained its stores: the DMA reads it back from L2), it does not fit next to
// img_head2's 110 KB of tiles.
template <class CH, class CM>
__global__ void __launch_bounds__(M3_NT)
img_fwd_all_kernel(const void* __restrict__ image, long bstride, const int* __restrict__ bidx, const float* __restrict__ wp1,
                   const float* __restrict__ wp2, float* __restrict__ y1, float* __restrict__ y2,
                   const float* __restrict__ wa3, const float* __restrict__ wa4, const float* __restrict__ wa5,
                   const float* __restrict__ params, int o_b1, int o_b2, int o_b3, int o_b4, int o_b5, int o_hb0, int o_hw1,
                   float* __restrict__ y3, float* __restrict__ y4, float* __restrict__ y5, const float* __restrict__ hw0t,
                   float* __restrict__ hid, float* __restrict__ part, int B, unsigned* sig, MidFinish fin) {
    static_assert(CH::NT == M3_NT && CH::LDS_FLOATS <= CM::W3A, "16 waves; conv 3's early filter half lies above the head's tiles");
    const int lane = threadIdx.x & 63;
    img_head2_body<CH>(image, bstride, bidx, wp1, params + o_b1, wp2, params + o_b2, y1, y2, B, (int)blockIdx.x, (int)gridDim.x,
                       [&](int wb) { mid3_early<CM>(wa3, params, o_b3, o_b4, o_b5, o_hb0, o_hw1, wb, lane); });
    // this workgroup's act2 stores have left the CU (every wave drains its own queue -- the early pieces with it), every wave is
    // done with the head's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    img_mid3_body<CM, true>(y2, wa3, wa4, wa5, params, o_b3, o_b4, o_b5, o_hb0, o_hw1, y3, y4, y5, hw0t, hid, part, blockIdx.x, sig, fin);
}
}  // namespace

// conv 1 .. conv 5 + image head at 84 x 84 as ONE launch (one image per workgroup: B <= 256); leaves act[1] (band-tiled), act[2..5],
// hid_i and the head partials like launch_img_fwd_head2 + launch_img_fwd_mid
int launch_img_fwd_all(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8, long bstride,
                       const int* image_index, int B) {
    using CM = M3Cfg<21>;
    ProfScope prof(c, s, TAG_IMG_FWD0 + 1);
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    auto go = [&](auto ch) -> int {
        using CH = decltype(ch);
        auto kern = img_fwd_all_kernel<CH, CM>;
        static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
        if (!(attr & var_dev_bit(c))) {
            VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CM::LDS_BYTES));
            attr |= var_dev_bit(c);
        }
        hipLaunchKernelGGL(kern, dim3(B), dim3(M3_NT), CM::LDS_BYTES, s, image, bstride, image_index, c->wpack + K.img_f[0],
                           c->wpack + K.img_f[1], c->act[1], c->act[2], c->wpack + K.img_f[2], c->wpack + K.img_f[3], c->wpack + K.img_f[4],
                           params, L.img_b[0], L.img_b[1], L.img_b[2], L.img_b[3], L.img_b[4], L.ih_b0, L.ih_w1, c->act[3], c->act[4],
                           c->act[5], c->wpack + K.ih_w0t, c->hid_i, c->head_part, B, c->dev_join ? c->jsig + 4 : nullptr,
                           MidFinish{c->mid_finish ? params + L.ih_b1 : nullptr, c->emb_raw, c->emb, c->out_img});
        VAR_HIP_CHECK(c, hipGetLastError());
        return VAR_OK;
    };
    return is_u8 ? go(H2_84u{}) : go(H2_84f{});
}

// conv 3 + conv 4 + conv 5 of the image CNN (act2 21 x 21 for 84 x 84 inputs, 24 x 24 for 96 x 96); leaves act[3], act[4],
// act[5] and, with_head, the image head's hidden layer (hid_i) and 128 -> 3 partials (head_part rows [0, B))
template <int H2>
static int launch_mid3(var_ctx* c, hipStream_t s, const float* params, int B, bool with_head) {
    using C = M3Cfg<H2>;
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_mid3_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr |= var_dev_bit(c);
    }
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    hipLaunchKernelGGL((img_mid3_kernel<C>), dim3(B), dim3(M3_NT), C::LDS_BYTES, s, c->act[2], c->wpack + K.img_f[2],
                       c->wpack + K.img_f[3], c->wpack + K.img_f[4], params, L.img_b[2], L.img_b[3], L.img_b[4], L.ih_b0, L.ih_w1,
                       c->act[3], c->act[4], c->act[5], with_head ? c->wpack + K.ih_w0t : nullptr, c->hid_i, c->head_part,
                       (with_head && c->dev_join) ? c->jsig + 4 : nullptr,
                       MidFinish{(with_head && c->mid_finish) ? params + L.ih_b1 : nullptr, c->emb_raw, c->emb, c->out_img});
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int launch_img_fwd_mid(var_ctx* c, hipStream_t s, const float* params, int B, bool with_head) {
    ProfScope prof(c, s, TAG_IMG_FWD0 + 2);
    return c->H == 84 ? launch_mid3<21>(c, s, params, B, with_head) : launch_mid3<24>(c, s, params, B, with_head);
}
