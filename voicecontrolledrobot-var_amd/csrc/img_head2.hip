// Head of the image CNN forward, second form: conv 1 (3 -> 32, u8 / f32 image) and conv 2 (32 -> 32), each
// Conv2d 3x3 stride 2 pad 1 + bias + ReLU (models/pretext/arm_pretext_model.py:9-12), for 84 x 84 and 96 x 96 inputs
// (96 x 96, the reference's img_dim in Envs/pybullet/arms/tasks/fourInARow/kuka/env_config.py:28: six bands of four act2 rows,
//  4 + 12 waves; 57.7 us alone at 256 images against the first form's 67.1, step 0.3618 -> 0.3538 ms on one box).
//
// Why a second form.  The first one (img_fwd_head.hip) is ONE 12-wave workgroup per CU that walks barrier-separated
// phases in lock step: while the conv-1 epilogue runs on the vector ALU nothing feeds the matrix cores and vice versa
// (phase timing in DESIGN.md: 52 % of the cycles have the matrix pipe busy).  This form is built so that the hardware
// overlaps the phases of DIFFERENT tiles by itself:
//   * small independent workgroups -- 4 waves, one per SIMD, 56 KB of LDS -- two per CU, each on its own tile (one band
//     of R2 act2 rows of one image); while one workgroup is in an epilogue / staging phase its SIMD partner multiplies;
//   * a wave is a complete conv engine: the WHOLE conv-2 filter lives in its registers as the A operand (144 VGPRs:
//     72 k-steps x 2 halves of the 32 output channels), so there is no K split, no fold through LDS, no filter traffic
//     in the tile loop, and one LDS read feeds two MFMAs;
//   * v_mfma_f32_16x16x4_f32 tiles (16 pixels x 16 channels x 4 k): 63 act2 pixels fill 4 pixel tiles to 98 %, one per
//     wave; two independent accumulators per wave cover the instruction's 40-cycle dependent latency;
//   * the staged image band and the act1 tile are stored column-parity split ([odd columns | even columns] per row), so
//     the stride-2 taps read consecutive floats (no 2-way bank conflicts on the B operand).
// Outputs: act2 (NCHW) and act1 -- band-tiled (84 x 84) or NCHW (96 x 96), see Head2Cfg; img_tail2.hip takes the ReLU mask of the
// backward from act1 itself (the ReLU bit image of rounds 1-3 is gone).
#include <stdlib.h>

#include <type_traits>

#include "var_common.h"

namespace {
typedef float f32x4h __attribute__((ext_vector_type(4)));

// TILED_: act1 leaves band-tiled (84 x 84: img_tail2.hip's bands are this kernel's); otherwise as NCHW rows (96 x 96: img_tail2.hip's
// twelve bands of four rows are gathered from them)
template <int H1_, bool U8_, int R2_, int NA_, bool TILED_>
struct Head2Cfg {
    static constexpr int H1 = H1_, W1 = H1_, R2 = R2_;
    static constexpr bool U8 = U8_, TILED = TILED_;
    static constexpr int CH = 32;
    static constexpr int HI = 2 * H1;                         // image 84
    static constexpr int HO2 = H1 / 2, WO2 = HO2;             // act2 plane 21
    static constexpr int NB = HO2 / R2;                       // bands per image
    static constexpr int IR1 = 2 * R2 + 1;                    // act1 rows per band
    static constexpr int IRI = 2 * IR1 + 1;                   // image rows per band
    // image band in LDS: [c][row][PWI], column = x + 4 (16-byte aligned groups of 4 pixels; column 3 = the zero pad x = -1)
    static constexpr int PWI = HI + 4;
    static constexpr int PLANE_I = IRI * PWI, IMG_FLOATS = 3 * PLANE_I;
    // act1 tile in LDS: per channel [2 pad][row 0 .. IR1-1, W1 floats each, no padding between rows][pad]: rows 1.. are one
    // contiguous 16-byte aligned run (= their run in HBM).  The cell x = -1 of a row is the previous row's last cell:
    // conv 2 zeroes that operand (lanes with ox = 0, taps kx = 0) instead of reading a pad.
    static constexpr int A1ROW0 = (W1 % 4 == 2) ? 2 : 4;
    static constexpr int PLANE_1 = ((A1ROW0 + IR1 * W1 + 7) / 8) * 8 + 4;  // = 4 mod 8: the epilogue's stores (4 q channels apart) fall on disjoint banks
    static constexpr int NPX1 = IR1 * W1, NT1 = (NPX1 + 15) / 16;          // conv-1 pixel tiles per band
    static constexpr int NPX2 = R2 * WO2, NT2 = (NPX2 + 15) / 16;          // conv-2 pixel tiles per band
    static constexpr int NA = NA_, NTA = 64 * NA;             // waves of role A (staging + conv 1): 8 (84 x 84) | 4 (96 x 96)
    static constexpr int NBW = 2 * NT2;                       // waves of role B (conv 2): 4 | 6 pixel tiles x 2 halves of the output channels
    static constexpr int NT = NTA + 64 * NBW;
    static constexpr int KS1 = 7;                             // conv 1: K = 27 -> 28 = 7 steps of 4
    static constexpr int KS2 = 9 * CH / 4;                    // conv 2: K = 288 = 72 steps of 4
    static constexpr int A1_FLOATS = CH * PLANE_1;
    static constexpr int IMS = 0;                             // two image-band buffers
    static constexpr int A1S = (2 * IMG_FLOATS + 3) & ~3;     // two act1-tile buffers
    static constexpr int LUT = A1S + 2 * A1_FLOATS;
    static constexpr int BIA = LUT + 256;
    static constexpr int LDS_FLOATS = BIA + 64;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int NLD = 3 * IRI * (HI / 4);            // 4-pixel groups of an image band
        static constexpr int LPT = (NLD + NTA - 1) / NTA;         // per lane of role A
    static_assert(HO2 % R2 == 0 && HI % 4 == 0, "whole bands, 4-pixel groups");
    static_assert(NT <= 1024 && NBW % 4 == 0 && NA % 4 == 0, "role B: one (conv-2 pixel tile, channel half) per wave; every role on every SIMD");
    static_assert(TILED || (2 * R2 * W1) % 4 == 0, "NCHW copy-out: a band's rows of a channel are whole float4");
    static_assert((A1ROW0 + W1) % 4 == 0 && PLANE_1 % 4 == 0 && (BIA % 4) == 0 && (IMG_FLOATS % 4) == 0 && PWI % 4 == 0, "16-byte aligned runs");
    static_assert(A1_FLOATS % 4 == 0, "the tile is whole float4");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

#ifndef VAR_HEAD2_DEVICE_ONLY
PH_DECL();
#ifdef VAR_PHASES
__device__ unsigned long long g_span_h2[1024][2];       // per workgroup: s_memrealtime (100 MHz, chip-wide) at its first and last instruction
#endif
#endif
}  // namespace
#if defined(VAR_PHASES) && !defined(VAR_HEAD2_DEVICE_ONLY)
extern "C" int var_debug_spans_head2(unsigned long long* out, int n) {      // tools/probe/head2_spans.py
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span_h2), sizeof(unsigned long long) * 2 * n) == hipSuccess ? 0 : -1;
}
extern "C" int var_debug_phases_head2(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
namespace {

// (a __device__ body: bx / G stand for blockIdx.x / gridDim.x of a stand-alone launch; `early` is run once by the waves of role B
//  before their first barrier -- img_fwd_all_kernel (img_mid3.hip) requests the next layers' filters there)
template <class C, class EARLY>
__device__ __forceinline__ void img_head2_body(const void* __restrict__ image, long bstride, const int* __restrict__ bidx,
                                               const float* __restrict__ wp1, const float* __restrict__ bias1,
                                               const float* __restrict__ wp2, const float* __restrict__ bias2,
                                               float* __restrict__ y1, float* __restrict__ y2, int B, const int bx, const int G,
                                               EARLY early) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using XT = typename std::conditional<C::U8, uint8_t, float>::type;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l15 = lane & 15;
    const float* lut = lds + C::LUT;
    // this workgroup's tiles: the NB bands of images blockIdx.x, blockIdx.x + gridDim.x, ...; tile j = (image, band)
    const int nimg = (bx < B) ? (B - 1 - bx) / G + 1 : 0;
    const int ntl = nimg * C::NB;
    auto tile_img = [&](int j) { return bx + (j / C::NB) * G; };

    PHR_INIT(3, VAR_PH_THREAD);
    if (tid < 256) lds[C::LUT + tid] = (float)tid / 255.f;
    if (tid < 32) { lds[C::BIA + tid] = bias1[tid]; lds[C::BIA + 32 + tid] = bias2[tid]; }
    for (int e = tid; e < 2 * 3 * C::IRI; e += C::NT) lds[C::IMS + e * C::PWI + 3] = 0.f;                     // pad column x = -1
    for (int e = tid; e < 2 * C::CH; e += C::NT) {              // pad cells of the act1 tiles (they travel to HBM with the tile)
        float* pl = lds + C::A1S + e * C::PLANE_1;
        for (int k = 0; k < C::A1ROW0; ++k) pl[k] = 0.f;
        for (int k = C::A1ROW0 + C::IR1 * C::W1; k < C::PLANE_1; ++k) pl[k] = 0.f;
    }
    __syncthreads();
    PHR(0);

    // act1 tile j (complete since the last barrier) -> HBM.  act1 lives in HBM band by band in EXACTLY the tile's LDS layout
    // ([image][band][channel][PLANE_1]: row 0 = the halo row the band shares with the one above, rows 1.. = the rows it
    // owns): one contiguous 38 KB block per band, copied 16 bytes per lane -- and read back the same way by the backward
    // kernel that is its only consumer (img_tail2.hip), which needs exactly these 2 R2 + 1 rows per band.  The NCHW form
    // (57.8 MB in 1 KB runs per channel and band) cost the forward 6 us and the backward its staging's index arithmetic.
    auto copy_out = [&](int j, int atid) {
        const int band = j % C::NB, b = tile_img(j);
        const float* a1 = lds + C::A1S + (j & 1) * C::A1_FLOATS;
        constexpr int NTC = 64 * C::NBW;                              // lanes of the role that runs the copy
        if constexpr (!C::TILED) {
            // NCHW: the band's 2 R2 owned rows of a channel are ONE run in the tile (rows 1.., no padding between rows) and in HBM
            constexpr int RUN4 = 2 * C::R2 * C::W1 / 4, N4 = C::CH * RUN4, NF = (N4 + NTC - 1) / NTC;
            float* yb = y1 + (size_t)b * C::CH * C::H1 * C::W1 + (size_t)band * 2 * C::R2 * C::W1;
            f32x4h cv[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                int e = atid + NTC * i;
                if (e >= N4) e = 0;
                const int ch = e / RUN4, o = e - ch * RUN4;
                cv[i] = *(const f32x4h*)(a1 + ch * C::PLANE_1 + C::A1ROW0 + C::W1 + 4 * o);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int e = atid + NTC * i;
                const int ch = e / RUN4, o = e - ch * RUN4;
                if (e < N4) *(f32x4h*)(yb + (size_t)ch * C::H1 * C::W1 + 4 * o) = cv[i];
            }
            return;
        }
        float* yb = y1 + (size_t)b * kAct1TiledFloats + (size_t)band * C::A1_FLOATS;
        // (the LDS reads of a group are issued before its first store: left alone hipcc emits read - wait - store chains)
        constexpr int NF4 = (C::A1_FLOATS / 4 + NTC - 1) / NTC;
        f32x4h cv[NF4];
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            int e = atid + NTC * i;
            if (e >= C::A1_FLOATS / 4) e = 0;
            cv[i] = *(const f32x4h*)(a1 + 4 * e);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int e = atid + NTC * i;
            if (e < C::A1_FLOATS / 4) *(f32x4h*)(yb + 4 * e) = cv[i];
        }
    };
    if (wave < C::NA) {
        // =====================================================================================================
        // role A: stage the image bands (u8 -> f32 through the table, column-parity split) and run conv 1
        // =====================================================================================================
        const int atid = tid;                                  // 0..255
        // Role A's matrix work comes in short bursts between vector-ALU epilogues, role B's is one long stream: with equal
        // priority the bursts are interleaved 1:1 with the stream and take twice as long, role A becomes the longer role
        // and the matrix pipe idles while role B waits at the barrier.  Served first, a burst runs at full rate and
        // role B fills every cycle role A leaves.
#ifndef VAR_H2_PRIO_A
#define VAR_H2_PRIO_A 2
#endif
        __builtin_amdgcn_s_setprio(VAR_H2_PRIO_A);
        // conv 1: A[i = l15][k = q] of k-step s; k = 4 s + q = tap*3 + c (row 27 of the packed filter is zero)
        float w1a[C::KS1], w1b[C::KS1];
        int koff[C::KS1];
#pragma unroll
        for (int s = 0; s < C::KS1; ++s) {
            const int k = 4 * s + q;
            w1a[s] = wp1[k * C::CH + l15];
            w1b[s] = wp1[k * C::CH + 16 + l15];
            const int kk = k > 26 ? 26 : k;
            const int tap = kk / 3, c = kk - tap * 3, ky = tap / 3, kx = tap - ky * 3;
            koff[s] = c * C::PLANE_I + ky * C::PWI + kx + 3;
        }
        // (the filter loads are waited for HERE: left to hipcc the s_waitcnt vmcnt(n) of each filter register sits in front of its
        //  first MFMA inside the tile loop, where -- the counter being in issue order -- it also waits for the copy-out's stores
        //  and the image prefetch issued just before)
#pragma unroll
        for (int s = 0; s < C::KS1; ++s) { asm volatile("" : "+v"(w1a[s])); asm volatile("" : "+v"(w1b[s])); }
        uint32_t ld_u8[C::LPT];
        float4 ld_f[C::U8 ? 1 : C::LPT];
        auto issue_band = [&](int j) {
            const int band = j % C::NB, b = tile_img(j);
            const int gi = bidx ? bidx[b] : b;
            const int iy0 = 4 * C::R2 * band - 3;
            const XT* im = (const XT*)image + (size_t)gi * bstride;
#pragma unroll
            for (int i = 0; i < C::LPT; ++i) {
                int e = atid + C::NTA * i;
                if (e >= C::NLD) e = 0;
                const int c = e / (C::IRI * (C::HI / 4)), rem = e - c * (C::IRI * (C::HI / 4));
                const int r = rem / (C::HI / 4), g = rem - r * (C::HI / 4);
                int iy = iy0 + r;
                iy = iy < 0 ? 0 : (iy >= C::HI ? C::HI - 1 : iy);
                const XT* src = im + (c * C::HI + iy) * C::HI + 4 * g;
                if constexpr (C::U8) ld_u8[i] = *(const uint32_t*)src;
                else ld_f[i] = *(const float4*)src;
            }
        };
        auto store_band = [&](int j) {
            const int band = j % C::NB;
            const int iy0 = 4 * C::R2 * band - 3;
            float* ims = lds + C::IMS + (j & 1) * C::IMG_FLOATS;
#pragma unroll
            for (int i = 0; i < C::LPT; ++i) {
                const int e = atid + C::NTA * i;
                if (e >= C::NLD) continue;
                const int c = e / (C::IRI * (C::HI / 4)), rem = e - c * (C::IRI * (C::HI / 4));
                const int r = rem / (C::HI / 4), g = rem - r * (C::HI / 4);
                const int iy = iy0 + r;
                const bool rok = iy >= 0 && iy < C::HI;
                float v0, v1, v2, v3;
                if constexpr (C::U8) {
                    const uint32_t w = rok ? ld_u8[i] : 0u;                 // lut[0] = 0
                    v0 = lut[w & 0xff]; v1 = lut[(w >> 8) & 0xff]; v2 = lut[(w >> 16) & 0xff]; v3 = lut[w >> 24];
                } else {
                    v0 = rok ? ld_f[i].x : 0.f; v1 = rok ? ld_f[i].y : 0.f; v2 = rok ? ld_f[i].z : 0.f; v3 = rok ? ld_f[i].w : 0.f;
                }
                *(float4*)(ims + c * C::PLANE_I + r * C::PWI + 4 + 4 * g) = make_float4(v0, v1, v2, v3);
            }
        };
        // conv 1 of tile j: image band in ims[j & 1] -> act1 tile a1s[j & 1] (bias + ReLU); copy_out() sends it on to HBM
        auto conv1 = [&](int j) {
            const int band = j % C::NB;
            const int imo = C::IMS + (j & 1) * C::IMG_FLOATS, a1o = C::A1S + (j & 1) * C::A1_FLOATS + C::A1ROW0;
            float bv[C::KS1];
            f32x4h c0, c1;
            auto fetch = [&](int t) {
                int p = t * 16 + l15;
                if (p >= C::NPX1) p = 0;
                const int rl = p / C::W1, x1 = p - rl * C::W1;
                const int pixbase = imo + 2 * rl * C::PWI + 2 * x1;
#pragma unroll
                for (int s = 0; s < C::KS1; ++s) bv[s] = lds[pixbase + koff[s]];
                c0 = *(const f32x4h*)(lds + C::BIA + 4 * q);              // the bias rides in as the accumulator's initial value
                c1 = *(const f32x4h*)(lds + C::BIA + 16 + 4 * q);
                if (band == 0 && t * 16 < C::W1) {                        // wave-uniform: the tile touches act1 row -1 (conv 2's zero padding):
                    const float ninf = -__builtin_inff();                 // -inf + finite products stays -inf, the ReLU makes it 0
                    if (p < C::W1) { c0 = {ninf, ninf, ninf, ninf}; c1 = c0; }
                }
            };
            fetch(wave);
#pragma unroll 1
            for (int t = wave; t < C::NT1; t += C::NA) {
#pragma unroll
                for (int s = 0; s < C::KS1; ++s) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[s], bv[s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1b[s], bv[s], c1, 0, 0, 0);
                }
                const f32x4h e0 = c0, e1 = c1;
                __builtin_amdgcn_sched_barrier(0);
                if (t + C::NA < C::NT1) fetch(t + C::NA);
                                 // next tile's operands: their LDS latency hides behind this epilogue
                const int p = t * 16 + l15;
                if (p < C::NPX1) {
                    float* d = lds + a1o + 4 * q * C::PLANE_1 + p;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d[r * C::PLANE_1] = __builtin_amdgcn_fmed3f(e0[r], 0.f, __builtin_inff());
                        d[(16 + r) * C::PLANE_1] = __builtin_amdgcn_fmed3f(e1[r], 0.f, __builtin_inff());
                    }
                }
            }
        };
        // ---- pipeline: step j runs conv 2 of tile j (role B) beside conv 1 of tile j + 1 and the staging of tile j + 2 ----
        if (ntl > 0) { issue_band(0); store_band(0); }
        if (ntl > 1) issue_band(1);
        __syncthreads();                                       // band 0 staged (4 waves of role A + the idle role B)
        if (ntl > 0) conv1(0);
        if (ntl > 1) store_band(1);
        PHR(1);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();                                   // act1 tile j complete, image band j + 1 staged
            PHR(2);
            if (j + 2 < ntl) issue_band(j + 2);                // in flight during conv 1
            PHR(5);
#ifndef VAR_H2_NOA
            if (j + 1 < ntl) conv1(j + 1);
#endif
            PHR(3);
            if (j + 2 < ntl) store_band(j + 2);                // into the buffer band j held: read for the last time in step j - 1
            PHR(4);
        }
        PHR_FLUSH();
    } else {
        // =====================================================================================================
        // role B: conv 2, one 16-pixel tile of the band per wave, all 32 output channels, the whole K = 288
        // =====================================================================================================
        const int wv = (wave - C::NA) % C::NT2, ct = (wave - C::NA) / C::NT2;
#ifdef VAR_H2_PRIO_B
        __builtin_amdgcn_s_setprio(VAR_H2_PRIO_B);
#endif
        // A[i = l15][k = q] of k-step s is Wf2[4 s + q][16 ct + l15] (Wf2[k][n], k = tap*32 + c): this wave's half of the filter
        float A0[C::KS2];
#pragma unroll
        for (int s = 0; s < C::KS2; ++s) A0[s] = wp2[(4 * s + q) * C::CH + 16 * ct + l15];
#pragma unroll
        for (int s = 0; s < C::KS2; ++s) asm volatile("" : "+v"(A0[s]));   // waited for here, see role A
        int p2 = wv * 16 + l15;
        const bool p2ok = p2 < C::NPX2;
        if (!p2ok) p2 = 0;
        const int oy2 = p2 / C::WO2, ox2 = p2 - oy2 * C::WO2;
        const int b2lane = q * C::PLANE_1 + C::A1ROW0 + 2 * oy2 * C::W1 + 2 * ox2 - 1;
        const bool edge = ox2 == 0;                              // taps kx = 0 of these lanes read the zero padding x = -1
        early(wave - C::NA);
        __syncthreads();                                       // (pairs with role A's first barrier)
        PHR(1);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();
            PHR(2);
            const int band = j % C::NB, b = tile_img(j);
            const int b2base = C::A1S + (j & 1) * C::A1_FLOATS + b2lane;
            f32x4h d0 = *(const f32x4h*)(lds + C::BIA + 32 + 16 * ct + 4 * q);       // bias as the initial value
            constexpr int UC = 4, NCK = C::KS2 / UC;             // operands are read one chunk of 4 k-steps ahead of their MFMAs
            float bb[2][UC];
            auto fetch = [&](int buf, int ck) {
                const int tap = (ck * UC) / 8, u0 = (ck * UC) % 8;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int to = ky * C::W1 + kx;
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const float v = lds[b2base + to + 4 * (u0 + u) * C::PLANE_1];
                    bb[buf][u] = (kx == 0 && edge) ? 0.f : v;
                }
            };
            fetch(0, 0);
#ifdef VAR_H2_NOB
            if (B < 0)
#endif
#pragma unroll
            for (int ck = 0; ck < NCK; ++ck) {
                if (ck + 1 < NCK) fetch((ck + 1) & 1, ck + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UC; ++u)
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[ck * UC + u], bb[ck & 1][u], d0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            PHR(3);
#ifndef VAR_HEAD2_NOKEEP
            copy_out(j, tid - C::NTA);                         // act1 tile j -> HBM: role B is the shorter role, and it is done with the matrix pipe here
#else
            (void)copy_out;
#endif
            PHR(5);
            if (p2ok) {
                float* yp = y2 + ((size_t)(b * C::CH + 16 * ct + 4 * q) * C::HO2 + band * C::R2) * C::WO2 + p2;
#pragma unroll
                for (int r = 0; r < 4; ++r) yp[(size_t)r * C::HO2 * C::WO2] = __builtin_amdgcn_fmed3f(d0[r], 0.f, __builtin_inff());
            }
            PHR(4);
        }
        PHR_FLUSH();
    }
}

template <class C>
__global__ void __launch_bounds__(C::NT)
img_head2_kernel(const void* __restrict__ image, long bstride, const int* __restrict__ bidx,
                 const float* __restrict__ wp1, const float* __restrict__ bias1,
                 const float* __restrict__ wp2, const float* __restrict__ bias2,
                 float* __restrict__ y1, float* __restrict__ y2, int B) {
#if defined(VAR_PHASES) && !defined(VAR_HEAD2_DEVICE_ONLY)
    if (threadIdx.x == 0) g_span_h2[blockIdx.x][0] = wall_clock64();
#endif
    img_head2_body<C>(image, bstride, bidx, wp1, bias1, wp2, bias2, y1, y2, B, (int)blockIdx.x, (int)gridDim.x, [](int) {});
#if defined(VAR_PHASES) && !defined(VAR_HEAD2_DEVICE_ONLY)
    if (threadIdx.x == 0) g_span_h2[blockIdx.x][1] = wall_clock64();
#endif
}

//                      H1   U8   R2 NA  TILED
using H2_84u = Head2Cfg<42, true, 3, 8, true>;
using H2_84f = Head2Cfg<42, false, 3, 8, true>;
// 96 x 96: six bands of four act2 rows (96 pixels = six full tiles: twelve waves of role B, three per SIMD, beside four of role A)
using H2_96u = Head2Cfg<48, true, 4, 4, false>;
using H2_96f = Head2Cfg<48, false, 4, 4, false>;

#ifndef VAR_HEAD2_DEVICE_ONLY      // img_mid3.hip includes this file for the device code above only
template <class C>
int launch_head2(var_ctx* c, hipStream_t s, const void* image, long bstride, const int* bidx, const float* params, int B) {
    ProfScope prof(c, s, TAG_IMG_FWD0 + 1);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_head2_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             C::LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const int G = B < kHead2G ? B : kHead2G;
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    hipLaunchKernelGGL(img_head2_kernel<C>, dim3(G), dim3(C::NT), C::LDS_BYTES, s, image, bstride, bidx,
                       c->wpack + K.img_f[0], params + L.img_b[0], c->wpack + K.img_f[1], params + L.img_b[1],
                       c->act[1], c->act[2], B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
#endif  // VAR_HEAD2_DEVICE_ONLY
}  // namespace

#ifndef VAR_HEAD2_DEVICE_ONLY
// var_debug_buffer("act1") at 84 x 84: the band-tiled act1 back to NCHW (rows 1.. of every band), into gact[1]
__global__ void __launch_bounds__(256) act1_untile_kernel(const float* __restrict__ t, float* __restrict__ y, long n) {
    using C = H2_84u;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % C::W1), r1 = (int)((i / C::W1) % C::H1), ch = (int)((i / (C::W1 * C::H1)) % C::CH);
    const long b = i / ((long)C::W1 * C::H1 * C::CH);
    const int band = r1 / (2 * C::R2), rl = r1 - band * 2 * C::R2 + 1;
    y[i] = t[b * kAct1TiledFloats + (long)band * C::A1_FLOATS + ch * C::PLANE_1 + C::A1ROW0 + rl * C::W1 + x];
}

int launch_act1_untile(var_ctx* c, hipStream_t s, int B) {
    const long n = (long)B * H2_84u::CH * H2_84u::H1 * H2_84u::W1;
    hipLaunchKernelGGL(act1_untile_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->act[1], c->gact[1], n);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// conv 1 + conv 2 of the image CNN; leaves act[2] and act[1] -- band-tiled at 84 x 84, NCHW at 96 x 96
int launch_img_fwd_head2(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8, long bstride,
                         const int* image_index, int B) {
    if (c->H == 96)
        return is_u8 ? launch_head2<H2_96u>(c, s, image, bstride, image_index, params, B)
                     : launch_head2<H2_96f>(c, s, image, bstride, image_index, params, B);
    return is_u8 ? launch_head2<H2_84u>(c, s, image, bstride, image_index, params, B)
                 : launch_head2<H2_84f>(c, s, image, bstride, image_index, params, B);
}
#endif  // VAR_HEAD2_DEVICE_ONLY
