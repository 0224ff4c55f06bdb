// End of the image CNN backward (84 x 84 and 96 x 96) in ONE role-specialised kernel: the weight gradient of conv 2, the data
// gradient of conv 2 and the weight gradient of conv 1 (autograd of models/pretext/arm_pretext_model.py:9-12 under
// loss.backward(), VAR/pretext_VAR.py:68).
//
// Round 2 ran this as two kinds of 12-wave workgroups in one grid (img_bwd_last_kernel: 256 split-K workgroups of the conv-2
// weight gradient, then 256 of the fused tail), each a chain of barrier-separated single-resource phases: 42 % of the f32
// matrix peak.  Here one persistent workgroup owns one image and walks its 7 bands (6 rows of gact1 each); per band the three
// products run SIDE BY SIDE in fixed wave roles, one wave of each role on every SIMD, so that the matrix pipe always has a
// taker while another role stages, masks or stores (the structure of img_head2.hip):
//   role D (4 waves)  data gradient of conv 2 on v_mfma_f32_16x16x4_f32: D[c][pixel] per column/row parity class, this wave's
//                     half of the transposed filter resident in 72 registers (no filter traffic, no K split, no fold), two
//                     pixel tiles per wave on independent accumulators; ReLU mask from the act1 band in LDS (the ReLU bit
//                     image of round 1-2 is gone); the masked band of gact1 goes to LDS only -- it never exists in HBM;
//   role W (4 waves)  weight gradient of conv 2: D[n][c] for the nine taps of one (16 n, 16 c) block per wave, reduction over
//                     the band's 63 output pixels; one gact2 operand read feeds nine MFMAs; accumulators live across bands;
//   role S (4 waves)  staging of the next band (act1 rows, gact2 rows, u8 image rows -> f32 through the 1/255 table) and the
//                     weight gradient of conv 1, D[n][(tap, c)], over the band of gact1 role D has just left in LDS.
// Both consumers of the act1 / gact2 bands read ONE staged copy (round 2 staged gact2 twice and read act1 with a 7/3 halo).
// Split-K over workgroups as before: one partial slab per workgroup and layer, summed in fixed order by
// img_wgrad_reduce_kernel (bitwise reproducible, no float atomics).
#include <stdlib.h>

#include <type_traits>

#include "var_common.h"

namespace {
PH_DECL();
}
#ifdef VAR_PHASES
extern "C" int var_debug_phases_tail2(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
namespace {
typedef float f32x4t __attribute__((ext_vector_type(4)));

// 84 x 84: H1 = 42, seven bands of RB = 6 gact1 rows, act1 band-tiled in HBM (img_head2.hip's tiles), 4 + 4 + 4 waves.
// 96 x 96: H1 = 48, twelve bands of RB = 4 rows (the 2 x 24 = 48 pixels of a parity class are three full tiles: six waves of role D,
// one tile each on two accumulators), act1 as NCHW rows (gathered by per-lane LDS-DMA addresses), 6 + 4 + 4 waves.
template <bool U8_, int H1_, int RB_, bool TILED_>
struct Tail2Cfg {
    static constexpr bool U8 = U8_, TILED = TILED_;
    static constexpr int CH = 32, H1 = H1_, W1 = H1_, HI = 2 * H1_, HO2 = H1_ / 2;
    static constexpr int RB = RB_, R2 = RB_ / 2, NB = H1_ / RB_;   // gact1 rows / act2 rows per band, bands per image
    // act1 band: EXACTLY img_head2.hip's tile, in LDS and (TILED) in HBM ([image][band][channel][A1_PLANE]): per channel A1_ROW0 pad
    // floats, then RB + 1 rows of W1 with no padding between them (row 0 = the halo row above the band).  The cell x = -1 of a row is
    // the previous row's last cell: role W zeroes that operand (pixels ox = 0, taps kx = 0) instead of reading a pad.
    static constexpr int A1_ROWS = RB + 1, A1_PITCH = W1, A1_ROW0 = (W1 % 4 == 2) ? 2 : 4;
    static constexpr int A1_PLANE = TILED ? ((A1_ROW0 + A1_ROWS * W1 + 7) / 8) * 8 + 4 : A1_ROW0 + A1_ROWS * W1;
    static constexpr int A1_FLOATS = CH * A1_PLANE;
    static constexpr int G2_ROWS = R2 + 1, G2_PITCH = HO2 + 1;                                                      // col = ox, col HO2 = 0
    static constexpr int G2_PLANE = ((G2_ROWS * G2_PITCH + 15) / 32) * 32 + 16, G2_FLOATS = CH * G2_PLANE;          // = 16 mod 32
    static constexpr int G1_PLANE = RB * W1 + ((RB * W1) % 32 == 0 ? 4 : 0), G1_FLOATS = CH * G1_PLANE;             // [c][RB x W1]
    static constexpr int IM_ROWS = 2 * RB + 1, IM_PITCH = HI + 4, IM_PLANE = IM_ROWS * IM_PITCH, IM_FLOATS = 3 * IM_PLANE;  // col = x + 4
    // PIPE (96 x 96): conv 1's weight gradient of band j - 1 (role S) runs beside the data gradient of band j (role D) -- two gact1 and
    // two image bands in LDS, ONE barrier per band; otherwise role D, which has no share of that product there, idles through it
    static constexpr bool PIPE = !TILED_;
    static constexpr int NBUF = PIPE ? 2 : 1;
    static constexpr int A1S = 0, G2S = A1S + 2 * A1_FLOATS, G1S = G2S + 2 * G2_FLOATS, IMS = G1S + NBUF * G1_FLOATS;
    static constexpr int LUT = IMS + NBUF * IM_FLOATS, ZCELL = LUT + 256;
    static constexpr int LDS_FLOATS = ZCELL + 4;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int NPX2 = R2 * HO2;                     // conv-2 output pixels of a band = pixels of a parity class of its gact1 rows (63 | 48)
    static constexpr int NT2 = (NPX2 + 15) / 16;              // their 16-pixel tiles (4 | 3)
    static constexpr int TPW = NT2 % 2 == 0 ? 2 : 1;          // tiles per wave of role D
    static constexpr int NACC = TPW == 1 ? 2 : 1;             // accumulators per tile (a lone tile alternates between two: the dependent latency)
    static constexpr int ND = 2 * NT2 / TPW;                  // waves of role D: (channel half, tile group) (4 | 6)
    static constexpr int NT = 64 * (ND + 8);                  // + 4 waves of role W + 4 of role S
    static constexpr int KS_W = (NPX2 + 3) / 4;               // k-steps of conv 2's weight gradient per band (16 | 12)
    static constexpr int NPX1 = RB * W1;                      // gact1 pixels of a band (252 | 192)
    static constexpr int KS_1 = NPX1 / 4;                     // k-steps of conv 1's weight gradient (63 | 48): [0, 4 SK) in role S, the rest in role D
    static constexpr int SK = KS_1 <= 48 ? KS_1 / 4 : 8;      // ... per wave of role S (8 | 12)
    static constexpr int DK = KS_1 <= 48 ? 0 : (((KS_1 - 32 + ND - 1) / ND + 3) / 4) * 4;   // ... per wave of role D (8 | none: at 14 waves its registers are the filter's)
    static constexpr int NSL = 4 + (DK ? ND : 0);             // K slices of conv 1's weight gradient to fold at the end
    static constexpr int N_A1 = A1_FLOATS / 4;                 // float4 of an act1 band
    static constexpr int P4 = A1_PLANE / 4;                   // ... of one channel's plane
    static constexpr int N_G2 = CH * G2_ROWS * HO2;           // floats of a gact2 band
    static constexpr int N_IM = 3 * IM_ROWS * (HI / 4);       // 4-pixel groups of an image band
    static constexpr int L_A1 = TILED ? (N_A1 + 255) / 256 : CH / 4;   /* wave instructions of the band's LDS-DMA per wave of role S (1 KiB chunks | one channel each) */
    static constexpr int L_G2 = (N_G2 + 255) / 256, L_IM = (N_IM + 255) / 256;
    static constexpr int SLAB0 = 32 * 32 + 32, SLAB1 = 32 * 9 * 32 + 32;
    static_assert(H1 % RB == 0 && RB % 2 == 0 && NPX1 % 4 == 0 && A1_PLANE % 4 == 0 && 4 * SK + DK * ND >= KS_1 && (SK == 8 || SK == 12) && (DK == 0 || DK == 8), "bands, k-step shares");
    static_assert(NT <= 1024 && LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(PIPE == (DK == 0), "the pipelined form is the one where role S holds all of conv 1's weight gradient");
    static_assert(NSL * 1024 + NSL * 32 <= 2 * A1_FLOATS, "the end-of-kernel fold scratch aliases the act1 buffers");
    static_assert(CH * HO2 * HO2 < (1 << 15) && CH * G2_PLANE < (1 << 12) && G2_ROWS <= 4, "role S: packed gact2 element table");
    static_assert(3 * HI * HI < (1 << 15) && 3 * IM_PLANE < (1 << 12) && IM_ROWS <= 16, "role S: packed image group table");
};

template <class C>
__global__ void __launch_bounds__(C::NT)
img_tail2_kernel(const float* __restrict__ gy, const float* __restrict__ wd, const float* __restrict__ act1,
                 const void* __restrict__ image, long bstride, const int* __restrict__ bidx,
                 float* __restrict__ slabs0, float* __restrict__ slabs1, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using XT = typename std::conditional<C::U8, uint8_t, float>::type;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l15 = lane & 15;
    const float* lut = lds + C::LUT;
    // this workgroup's tiles: the NB bands of images blockIdx.x, blockIdx.x + gridDim.x, ...
    const int nimg = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ntl = nimg * C::NB;
    auto tile_img = [&](int j) { return (int)blockIdx.x + (j / C::NB) * (int)gridDim.x; };

    PHR_INIT(3, VAR_PH_THREAD);
    if (tid < 256) lds[C::LUT + tid] = (float)tid / 255.f;
    if (tid < 4) lds[C::ZCELL + tid] = 0.f;
    for (int e = tid; e < 2 * C::G2_FLOATS; e += C::NT) lds[C::G2S + e] = 0.f;                                    // (incl. the zero column ox = 21)
    for (int e = tid; e < C::NBUF * 3 * C::IM_ROWS; e += C::NT) lds[C::IMS + e * C::IM_PITCH + 3] = 0.f;           // x = -1
    __syncthreads();                                           // the pad cells are in place before role S stages band 0

    float* slab0 = slabs0 + (size_t)blockIdx.x * C::SLAB0;
    float* slab1 = slabs1 + (size_t)blockIdx.x * C::SLAB1;
    f32x4t s_acc[2][2];                                       // role S: conv 1's weight gradient, [n tile][column tile]
    float s_bs[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) s_acc[a][b] = {0.f, 0.f, 0.f, 0.f};

    // ---- conv 1's weight gradient over the band of gact1 in LDS (roles S and D share it: each wave one chunk of 8 k-steps) ----
    // columns (tap, c) = 16 kt + l15 (27 of 32 used; the rest read column 26 and are dropped)
    int cbase[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        int col = 16 * kt + l15;
        col = col > 26 ? 26 : col;
        const int tap = col / 3, c = col - tap * 3;
        cbase[kt] = C::IMS + c * C::IM_PLANE + (tap / 3) * C::IM_PITCH + (tap % 3) + 3;
    }
    // k-step s: pixels m = 4 s + q of the band -> (row yl, column x); gact1 at [n][m], image at rows 2 yl + ky, columns
    // 2 x + kx - 1.  The operands of the chunk's 8 k-steps are read ahead of its 32 MFMAs.
    // (opq: 0, opaque to the compiler where the caller wants the chunk's twelve addresses recomputed per band instead of hoisted out of
    //  the band loop into registers)
    auto wgrad1_chunk = [&](auto ck, int s_lo, int s_hi, int opq, int g1o = 0, int imo = 0) {       // (g1o / imo: the band's gact1 / image buffer, PIPE)
        int m = 4 * s_lo + q + opq;
        int yl = m / C::W1, x = m - yl * C::W1;
        const int ga = C::G1S + g1o + l15 * C::G1_PLANE;
        constexpr int CK = decltype(ck)::value;
        float a0[CK], a1v[CK], b0[CK], b1[CK];
#pragma unroll
        for (int u = 0; u < CK; ++u) {
            const bool live = s_lo + u < s_hi;                 // (wave-uniform; dead steps multiply zeros)
            const int pix = 2 * yl * C::IM_PITCH + 2 * x;
            // (selects on the ADDRESS, loads unconditional: a select on the loaded value becomes a branch around the load)
            const int ao = live ? ga + m : C::ZCELL, po = live ? pix : 0;
            a0[u] = lds[ao];
            a1v[u] = lds[live ? ao + 16 * C::G1_PLANE : C::ZCELL];
            b0[u] = lds[cbase[0] + imo + po];
            b1[u] = lds[cbase[1] + imo + po];
            m += 4; x += 4;
            if (x >= C::W1) { x -= C::W1; ++yl; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CK; ++u) {
            s_bs[0] += a0[u]; s_bs[1] += a1v[u];
            s_acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], s_acc[0][0], 0, 0, 0);
            s_acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b1[u], s_acc[0][1], 0, 0, 0);
            s_acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[u], b0[u], s_acc[1][0], 0, 0, 0);
            s_acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[u], b1[u], s_acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // The gact2 rows of a band, global -> registers -> LDS, by NL lanes numbered lt (role S's 256; at 96 x 96 role D's 384: role S is the
    // longest role there, role D waits).  Element e = (channel, row r of the band, ox); a row below the image reads the row above and
    // stores zero.
    // (a lane keeps its place (row, ox) in a channel's G2_ROWS x HO2 block and walks the channels CPP at a time: per pass one add for the
    //  global and one for the LDS offset -- element-wise index arithmetic made these few loads the longest stretch of the role)
    auto g2_load = [&](auto nl, float* r, int j, int lt) {
        constexpr int NL = decltype(nl)::value, WIN = C::G2_ROWS * C::HO2, CPP = NL / WIN, L = (C::CH + CPP - 1) / CPP;
        const int band = j % C::NB, b = tile_img(j);
        const int cs = lt / WIN, w = lt - cs * WIN, rr = w / C::HO2;
        const int back = (C::R2 * band + rr >= C::HO2) ? C::HO2 : 0;
        const float* pg = gy + (size_t)b * C::CH * C::HO2 * C::HO2 + C::R2 * band * C::HO2 + cs * C::HO2 * C::HO2 + w - back;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const bool ok = cs < CPP && cs + CPP * i < C::CH;
            r[i] = pg[ok ? CPP * i * C::HO2 * C::HO2 : 0];
        }
    };
    auto g2_store = [&](auto nl, const float* r, int j, int lt) {
        constexpr int NL = decltype(nl)::value, WIN = C::G2_ROWS * C::HO2, CPP = NL / WIN, L = (C::CH + CPP - 1) / CPP;
        const int band = j % C::NB;
        const int cs = lt / WIN, w = lt - cs * WIN, rr = w / C::HO2, ox = w - rr * C::HO2;
        float* dg = lds + C::G2S + (j & 1) * C::G2_FLOATS + cs * C::G2_PLANE + rr * C::G2_PITCH + ox;
        const bool live = C::R2 * band + rr < C::HO2;
#pragma unroll
        for (int i = 0; i < L; ++i)
            if (cs < CPP && cs + CPP * i < C::CH) dg[CPP * i * C::G2_PLANE] = live ? r[i] : 0.f;
    };
    constexpr bool G2_IN_D = C::DK == 0;                      // (96 x 96)
    constexpr int NLD_ = 64 * C::ND, L_G2D = (C::CH + NLD_ / (C::G2_ROWS * C::HO2) - 1) / (NLD_ / (C::G2_ROWS * C::HO2));

    // a wave's K slice of conv 1's weight gradient -> LDS (the act1 buffers, dead by then: behind a workgroup barrier)
    auto park_slices = [&](int sl) {                           // slice = its range of k-steps, ascending
        float* d = lds + sl * 1024;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[(16 * a + 4 * q + r) * 32 + 16 * b + l15] = s_acc[a][b][r];
        // bias: lane (q, l15) summed its k index q of every step; fold q, then park per wave
        float b0 = s_bs[0], b1 = s_bs[1];
        b0 += __shfl_xor(b0, 16, 64); b0 += __shfl_xor(b0, 32, 64);
        b1 += __shfl_xor(b1, 16, 64); b1 += __shfl_xor(b1, 32, 64);
        if (q == 0) { lds[C::NSL * 1024 + sl * 32 + l15] = b0; lds[C::NSL * 1024 + sl * 32 + 16 + l15] = b1; }
    };

    if (wave < C::ND) {
        // =====================================================================================================
        // role D: data gradient of conv 2 for this wave's two pixel tiles and channel half, masked, into LDS
        // =====================================================================================================
        const int ct = wave & 1, pp = wave >> 1;
        float A[72];                                           // A[i = c][k = n] of (tap, k-step s): Wd[tap][4 s + q][16 ct + l15]
#pragma unroll
        for (int t = 0; t < 72; ++t) A[t] = wd[((t >> 3) * 32 + 4 * (t & 7) + q) * 32 + 16 * ct + l15];
#pragma unroll
        for (int t = 0; t < 72; ++t) asm volatile("" : "+v"(A[t]));      // waited for here, not inside the band loop (img_head2.hip)
        int mj[C::TPW], mi[C::TPW];
        bool mok[C::TPW];
#pragma unroll
        for (int u = 0; u < C::TPW; ++u) {
            int m = (C::TPW * pp + u) * 16 + l15;                   // pixel of the parity class: (row pair j, column pair i)
            mok[u] = m < C::NPX2;
            if (!mok[u]) m = 0;
            mj[u] = m / C::HO2; mi[u] = m - mj[u] * C::HO2;
        }
        __syncthreads();                                       // (P) band 0 staged
        PHR(0);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();                                   // (A)
            PHR(1);
            float r_g2d[G2_IN_D ? L_G2D : 1];
            int dtid = tid;
            if constexpr (G2_IN_D) {
                asm volatile("" : "+v"(dtid));                   // (opaque per band: the element math is not to be hoisted into registers)
                if (j + 1 < ntl) g2_load(std::integral_constant<int, NLD_>{}, r_g2d, j + 1, dtid);
            }
            const int g2 = C::G2S + (j & 1) * C::G2_FLOATS + q * C::G2_PLANE;
            const int a1 = C::A1S + (j & 1) * C::A1_FLOATS;
            // nine (parity class, tap) chunks of 8 k-steps x 2 pixel tiles; the operands of chunk k + 1 are read before the 16
            // MFMAs of chunk k (pinned with scheduling barriers: left alone hipcc sinks every read to its use and each pair of
            // MFMAs then waits out an LDS latency); a class's accumulators are masked and stored behind its last chunk
            constexpr int kCls[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};                   // class = 2 py + px
            constexpr int kKy[9] = {1, 1, 1, 0, 2, 0, 0, 2, 2}, kKx[9] = {1, 0, 2, 1, 1, 0, 2, 0, 2};
            // (opaque per band: hipcc otherwise hoists the sixteen mask / store addresses of the epilogues out of the band loop)
            int mjb[C::TPW], mib[C::TPW], lb[C::TPW];
#pragma unroll
            for (int u = 0; u < C::TPW; ++u) {
                mjb[u] = mj[u]; mib[u] = mi[u];
                asm volatile("" : "+v"(mjb[u])); asm volatile("" : "+v"(mib[u]));
                lb[u] = g2 + mjb[u] * C::G2_PITCH + mib[u];
            }
            // (operands in half chunks of 4 k-steps: 16 registers in flight beside the filter's 72)
            float bb[2][C::TPW][4];
            auto fetch = [&](int buf, int h) {                 // half chunk h = 2 k + (upper half)
                const int k = h >> 1, s0 = 4 * (h & 1);
                // gy row j + 1 for ky = 0 (odd rows), column i + 1 for kx = 0 (odd columns)
                const int off = (kKy[k] == 0 ? C::G2_PITCH : 0) + (kKx[k] == 0 ? 1 : 0);
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    bb[buf][0][ss] = lds[lb[0] + off + 4 * (s0 + ss) * C::G2_PLANE];
                    if constexpr (C::TPW == 2) bb[buf][C::TPW - 1][ss] = lds[lb[C::TPW - 1] + off + 4 * (s0 + ss) * C::G2_PLANE];
                }
            };
            f32x4t acc[C::TPW * C::NACC];
#pragma unroll
            for (int u = 0; u < C::TPW * C::NACC; ++u) acc[u] = {0.f, 0.f, 0.f, 0.f};
            fetch(0, 0);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                const int k = h >> 1;
                if (h + 1 < 18) fetch((h + 1) & 1, h + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int tap = kKy[k] * 3 + kKx[k];
#pragma unroll
                for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                    for (int u = 0; u < C::TPW; ++u)
                        acc[u * C::NACC + ss % C::NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tap * 8 + 4 * (h & 1) + ss], bb[h & 1][u][ss],
                                                                                    acc[u * C::NACC + ss % C::NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(h & 1)) continue;
                if (k + 1 == 9 || kCls[k + 1] != kCls[k]) {
                    // ReLU mask of act1 at (row 2 j + py, column 2 i + px), then the band of gact1 in LDS
                    const int py = kCls[k] >> 1, px = kCls[k] & 1;
#pragma unroll
                    for (int u = 0; u < C::TPW; ++u) {
                        f32x4t tot = acc[u * C::NACC];
                        if constexpr (C::NACC == 2) tot += acc[u * C::NACC + 1];
                        const int yl = 2 * mjb[u] + py, x = 2 * mib[u] + px;
                        const int mo = a1 + (16 * ct + 4 * q) * C::A1_PLANE + C::A1_ROW0 + (yl + 1) * C::A1_PITCH + x;
                        const int go = C::G1S + (C::PIPE ? (j & 1) * C::G1_FLOATS : 0) + (16 * ct + 4 * q) * C::G1_PLANE + yl * C::W1 + x;
                        float mv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) mv[r] = lds[mo + r * C::A1_PLANE];
                        if (mok[u]) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) lds[go + r * C::G1_PLANE] = mv[r] > 0.f ? tot[r] : 0.f;
                        }
#pragma unroll
                        for (int a = 0; a < C::NACC; ++a) acc[u * C::NACC + a] = {0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            if constexpr (G2_IN_D) {
                asm volatile("" : "+v"(dtid));
                if (j + 1 < ntl) g2_store(std::integral_constant<int, NLD_>{}, r_g2d, j + 1, dtid);   // (the other gact2 buffer: nobody reads it in this band)
            }
            PHR(2);
            if constexpr (!C::PIPE) __syncthreads();           // (B) the band of gact1 is complete
            PHR(3);
            // conv 1's weight gradient: this wave's DK k-steps, in chunks of 4 (the filter's 72 registers leave no room for 8)
            if constexpr (C::DK == 8) {
                wgrad1_chunk(std::integral_constant<int, 4>{}, 32 + 8 * wave, 32 + 8 * wave + 4, 0);
                wgrad1_chunk(std::integral_constant<int, 4>{}, 32 + 8 * wave + 4, 32 + 8 * wave + 8 < C::KS_1 ? 32 + 8 * wave + 8 : C::KS_1, 0);
            }
            PHR(4);
        }
        PHR_FLUSH();
        if constexpr (C::PIPE) __syncthreads();             // (the last band is complete: role S still has its conv-1 weight gradient to do)
        if constexpr (C::DK == 0) __syncthreads();          // (the fold's first barrier: see below)
    } else if (wave < C::ND + 4) {
        // =====================================================================================================
        // role W: weight gradient of conv 2, the nine taps of this wave's (16 n, 16 c) block, summed over all bands
        // =====================================================================================================
        const int w = wave - C::ND, nt = w & 1, ctl = w >> 1;
        int aoff[C::KS_W], boff[C::KS_W];
        uint32_t edge = 0;
#pragma unroll
        for (int s = 0; s < C::KS_W; ++s) {
            const int m = 4 * s + q;                           // output pixel of the band: (row r, column ox)
            const bool ok = m < C::NPX2;
            const int mm = ok ? m : 0;
            const int r = mm / C::HO2, ox = mm - r * C::HO2;
            aoff[s] = ok ? (16 * nt + l15) * C::G2_PLANE + r * C::G2_PITCH + ox : -1;
            boff[s] = (16 * ctl + l15) * C::A1_PLANE + C::A1_ROW0 + 2 * r * C::A1_PITCH + 2 * ox - 1;     // + ky * PITCH + kx
            if (ox == 0) edge |= 1u << s;                     // taps kx = 0 of this pixel read the zero padding x = -1
        }
        f32x4t acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
        __syncthreads();                                       // (P)
        PHR(0);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();                                   // (A)
            PHR(1);
            const int g2 = C::G2S + (j & 1) * C::G2_FLOATS, a1 = C::A1S + (j & 1) * C::A1_FLOATS;
            // per k-step one gact2 operand (A) and the nine tap-shifted act1 operands (B); those of step s + 1 are read before
            // the nine MFMAs of step s
            float av[2], bv[2][9];
            auto fetch = [&](int buf, int ss) {
                av[buf] = lds[aoff[ss] < 0 ? C::ZCELL : g2 + aoff[ss]];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = lds[a1 + boff[ss] + (t / 3) * C::A1_PITCH + (t % 3)];
                    bv[buf][t] = (t % 3 == 0 && ((edge >> ss) & 1)) ? 0.f : v;
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int ss = 0; ss < C::KS_W; ++ss) {
                if (ss + 1 < C::KS_W) fetch((ss + 1) & 1, ss + 1);
                __builtin_amdgcn_sched_barrier(0);
                bsum += av[ss & 1];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ss & 1], bv[ss & 1][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ss + 1 == C::KS_W / 2) { PHR(2); if constexpr (!C::PIPE) __syncthreads(); PHR(3); }   // (B): reads only, either side of it
            }
            PHR(4);
            // images change every NB bands: nothing to do, the accumulators run on
        }
        // this workgroup's partial slab of layer 1: [n][tap][c] + bias sums
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab1[((16 * nt + 4 * q + r) * 9 + t) * 32 + 16 * ctl + l15] = acc[t][r];
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (ctl == 0 && q == 0) slab1[9216 + 16 * nt + l15] = bsum;
        PHR_FLUSH();
        if constexpr (C::PIPE) __syncthreads();
        if constexpr (C::DK == 0) __syncthreads();
    } else {
        // =====================================================================================================
        // role S: staging of the next band + weight gradient of conv 1 over the band of gact1 in LDS
        // =====================================================================================================
        const int sw = wave - C::ND - 4, stid0 = tid - 64 * (C::ND + 4);
        float r_g2[C::L_G2];
        uint32_t r_u8[C::U8 ? C::L_IM : 1];
        float4 r_f4[C::U8 ? 1 : C::L_IM];
        // per-lane element tables, computed once (they do not depend on the band): gact2 element e -> (global offset inside the
        // image's band rows | LDS offset | row r) packed in one register; image 4-pixel group e -> (global | LDS | row) likewise
        // (96 x 96: no tables, the packed entry is recomputed where it is used -- at 14 waves a wave has 128 registers)
        auto make_g2 = [&](int e) -> uint32_t {
            const int ee = e < C::N_G2 ? e : 0;
            const int ch = ee / (C::G2_ROWS * C::HO2), rem = ee - ch * (C::G2_ROWS * C::HO2);
            const int r = rem / C::HO2, ox = rem - r * C::HO2;
            // global: ch * HO2^2 + r * HO2 + ox (< 2^15); LDS: ch * G2_PLANE + r * G2_PITCH + ox (< 2^12); row r (< 4); bit 31 = no element
            return (uint32_t)(ch * C::HO2 * C::HO2 + r * C::HO2 + ox) | ((uint32_t)(ch * C::G2_PLANE + r * C::G2_PITCH + ox) << 15) |
                   ((uint32_t)r << 27) | (e < C::N_G2 ? 0u : 0x80000000u);
        };
        auto make_im = [&](int e) -> uint32_t {
            const int ee = e < C::N_IM ? e : 0;
            const int c = ee / (C::IM_ROWS * (C::HI / 4)), rem = ee - c * (C::IM_ROWS * (C::HI / 4));
            const int r = rem / (C::HI / 4), g = rem - r * (C::HI / 4);
            // global: c * HI * HI + r * HI + 4 g (< 2^15); LDS: c * IM_PLANE + r * IM_PITCH + 4 + 4 g (< 2^12); row r (< 16)
            return (uint32_t)(c * C::HI * C::HI + r * C::HI + 4 * g) | ((uint32_t)(c * C::IM_PLANE + r * C::IM_PITCH + 4 + 4 * g) << 15) |
                   ((uint32_t)r << 27) | (e < C::N_IM ? 0u : 0x80000000u);
        };
        constexpr bool TAB = C::DK != 0;
        uint32_t t_g2[TAB ? C::L_G2 : 1], t_im[TAB ? C::L_IM : 1];
        if constexpr (TAB) {
#pragma unroll
            for (int i = 0; i < C::L_G2; ++i) t_g2[i] = make_g2(stid0 + 256 * i);
#pragma unroll
            for (int i = 0; i < C::L_IM; ++i) t_im[i] = make_im(stid0 + 256 * i);
        }
        auto e_g2 = [&](int i, int stid) -> uint32_t { if constexpr (TAB) return t_g2[i]; else return make_g2(stid + 256 * i); };
        // (96 x 96: lane = (row, 4-pixel group) of a channel's rows, pass i = channel i: one entry per lane, recomputed per band)
        static_assert(TAB || (C::L_IM == 3 && C::IM_ROWS * (C::HI / 4) <= 256), "image band: a channel per pass");
        auto e_im = [&](int i, int stid) -> uint32_t {
            if constexpr (TAB) return t_im[i];
            else return stid < C::IM_ROWS * (C::HI / 4) ? make_im(stid) : 0x80000000u;
        };
        auto issue_a1g2 = [&](int j, int stid, bool with_g2 = true) {
            const int band = j % C::NB, b = tile_img(j);
            // the act1 band: ONE contiguous block in HBM in the tile's own layout (img_head2.hip) -> straight into LDS by
            // LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes land at the wave's base + 16 l), 1 KiB per wave instruction,
            // no registers, no store pass; it lands during this band's matrix work and is waited for before the next barrier (A)
            // NCHW (96 x 96): the same tile gathered -- a lane's 16 bytes are four floats of one channel's run of RB + 1 rows (plus the pad
            // cells in front of it: whatever lies there), at its own address; the halo row above band 0 does not exist (read: row 0's
            // neighbourhood, clamped into the image) and is zeroed behind the copy (zero_halo)
            const float* pa = C::TILED ? act1 + (size_t)b * kAct1TiledFloats + (size_t)band * C::A1_FLOATS
                                       : act1 + (size_t)b * C::CH * C::H1 * C::W1;
            float* da = lds + C::A1S + (j & 1) * C::A1_FLOATS;
#pragma unroll
            for (int i = 0; i < C::L_A1; ++i) {
                const int k = sw + 4 * i;                                              // 1 KiB chunk of this wave
                if constexpr (C::TILED) {
                    if (k * 64 + lane < C::N_A1)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + k * 256 + 4 * lane),
                                                         (__attribute__((address_space(3))) void*)(da + k * 256), 16, 0, 0);
                } else {
                    // one channel's plane (P4 float4: the pad cells + RB + 1 rows) per wave instruction, channel k; lanes >= P4 idle
                    static_assert(C::TILED || (C::P4 <= 64 && C::L_A1 * 4 == C::CH), "NCHW gather: a channel per instruction");
                    int off = k * C::H1 * C::W1 + (C::RB * band - 1) * C::W1 - C::A1_ROW0 + 4 * lane;
                    off = off < 0 ? 0 : off;
                    if (lane < C::P4)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + off),
                                                         (__attribute__((address_space(3))) void*)(da + k * C::A1_PLANE), 16, 0, 0);
                }
            }
            if (!with_g2) return;
            const float* pg = gy + (size_t)b * C::CH * C::HO2 * C::HO2 + C::R2 * band * C::HO2;
#pragma unroll
            for (int i = 0; i < C::L_G2; ++i) {
                const uint32_t tg = e_g2(i, stid);
                const int r = (tg >> 27) & 3;
                const int back = (C::R2 * band + r >= C::HO2) ? C::HO2 : 0;           // a row below the image: read the row above (discarded)
                r_g2[i] = pg[(int)(tg & 0x7fff) - back];
            }
        };
        auto store_a1g2 = [&](int j, int stid) {
            const int band = j % C::NB;
            float* dg = lds + C::G2S + (j & 1) * C::G2_FLOATS;
#pragma unroll
            for (int i = 0; i < C::L_G2; ++i) {
                const uint32_t tg = e_g2(i, stid);
                const int r = (tg >> 27) & 3;
                if (!(tg >> 31)) dg[(tg >> 15) & 0xfff] = (C::R2 * band + r < C::HO2) ? r_g2[i] : 0.f;
            }
        };
        auto issue_img = [&](int j, int stid) {
            const int band = j % C::NB, b = tile_img(j);
            const int gi = bidx ? bidx[b] : b;
            const int iy0 = 2 * C::RB * band - 1;
            const XT* im = (const XT*)image + (size_t)gi * bstride;
#pragma unroll
            for (int i = 0; i < C::L_IM; ++i) {
                const uint32_t ti = e_im(i, stid);
                const int r = (ti >> 27) & 15;
                const int iy = iy0 + r;
                const int fix = iy < 0 ? C::HI : 0;                                   // row -1: read row 0 (discarded)
                const XT* src = im + (int)(ti & 0x7fff) + iy0 * C::HI + fix;
                if constexpr (!TAB) src = im + (int)(ti & 0x7fff) + i * C::HI * C::HI + iy0 * C::HI + fix;
                if constexpr (C::U8) r_u8[i] = *(const uint32_t*)src;
                else r_f4[i] = *(const float4*)src;
            }
        };
        auto store_img = [&](int j, int stid) {
            const int band = j % C::NB;
            const int iy0 = 2 * C::RB * band - 1;
#pragma unroll
            for (int i = 0; i < C::L_IM; ++i) {
                const uint32_t ti = e_im(i, stid);
                if (ti >> 31) continue;
                const int r = (ti >> 27) & 15;
                const bool rok = iy0 + r >= 0;                                        // (rows >= HI do not occur: 12 * 6 + 11 = 83)
                float4 v;
                if constexpr (C::U8) {
                    const uint32_t wv = rok ? r_u8[i] : 0u;             // lut[0] = 0
                    v = make_float4(lut[wv & 0xff], lut[(wv >> 8) & 0xff], lut[(wv >> 16) & 0xff], lut[wv >> 24]);
                } else {
                    v = rok ? r_f4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *(float4*)(lds + C::IMS + (C::PIPE ? (j & 1) * C::IM_FLOATS : 0) + ((ti >> 15) & 0xfff) + (TAB ? 0 : i * C::IM_PLANE)) = v;
            }
        };
        // NCHW: the halo row above band 0 (act1 row -1 = conv 2's zero padding), behind the band's LDS-DMA
        auto zero_halo = [&](int j, int stid) {
            if constexpr (!C::TILED) {
                if (j % C::NB == 0) {
                    float* da = lds + C::A1S + (j & 1) * C::A1_FLOATS + C::A1_ROW0;
                    for (int e = stid; e < C::CH * (C::W1 / 4); e += 256) {
                        const int ch = e / (C::W1 / 4), o = e - ch * (C::W1 / 4);
                        *(float4*)(da + ch * C::A1_PLANE + 4 * o) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        };
        if (ntl > 0) { issue_a1g2(0, stid0); issue_img(0, stid0); store_a1g2(0, stid0); }     // (every request of band 0 out before the first wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ntl > 0) zero_halo(0, stid0);
        __syncthreads();                                       // (P)
        PHR(0);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();                                   // (A)
            PHR(1);
            // (an opaque copy of the lane id per band: hipcc otherwise hoists every element's index math out of the band
            //  loop and keeps ~150 registers of it alive -- or in scratch -- for the life of the kernel)
            int stid = stid0;
            asm volatile("" : "+v"(stid));
            store_img(j, stid);
            if constexpr (C::DK != 0) {
                if (j + 1 < ntl) { issue_img(j + 1, stid); issue_a1g2(j + 1, stid); }   // a whole band of matrix work ahead of their use
                PHR(2);
                __syncthreads();                               // (B) gact1 band complete (role D), image band stored (this role)
                PHR(3);
                int z8 = 0;
                asm volatile("" : "+v"(z8));
                wgrad1_chunk(std::integral_constant<int, 8>{}, 8 * sw, 8 * sw + 8, z8);    // k-steps [0, 32) here, the rest in role D
                PHR(4);
                asm volatile("" : "+v"(stid));
                if (j + 1 < ntl) store_a1g2(j + 1, stid);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the band's LDS-DMA has landed (and, with it, the image loads)
            } else {
                // 96 x 96 (PIPE): the next band's act1 by LDS-DMA and image rows, then ALL of conv 1's weight gradient of the PREVIOUS band
                // (its gact1 and image bands stay in their buffers for one more step); the gact2 rows are role D's here
                if (j + 1 < ntl) { issue_img(j + 1, stid); issue_a1g2(j + 1, stid, false); }
                PHR(2);
                PHR(3);
                if (j > 0) {
                    int z = 0;
                    asm volatile("" : "+v"(z));
                    const int g1o = ((j - 1) & 1) * C::G1_FLOATS, imo = ((j - 1) & 1) * C::IM_FLOATS;
                    wgrad1_chunk(std::integral_constant<int, 4>{}, C::SK * sw, C::SK * sw + 4, z, g1o, imo);      // k-steps [0, 4 SK): all of them
                    wgrad1_chunk(std::integral_constant<int, 4>{}, C::SK * sw + 4, C::SK * sw + 8, z, g1o, imo);
                    wgrad1_chunk(std::integral_constant<int, 4>{}, C::SK * sw + 8, C::SK * sw + 12, z, g1o, imo);
                }
                PHR(4);
                asm volatile("" : "+v"(stid));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the band's LDS-DMA has landed (and, with it, the image loads)
            }
            if (j + 1 < ntl) zero_halo(j + 1, stid);
            PHR(5);
        }
        PHR_FLUSH();
        if constexpr (C::PIPE) {
            __syncthreads();
            if (ntl > 0) {
                const int g1o = ((ntl - 1) & 1) * C::G1_FLOATS, imo = ((ntl - 1) & 1) * C::IM_FLOATS;
                wgrad1_chunk(std::integral_constant<int, 4>{}, C::SK * sw, C::SK * sw + 4, 0, g1o, imo);
                wgrad1_chunk(std::integral_constant<int, 4>{}, C::SK * sw + 4, C::SK * sw + 8, 0, g1o, imo);
                wgrad1_chunk(std::integral_constant<int, 4>{}, C::SK * sw + 8, C::SK * sw + 12, 0, g1o, imo);
            }
        }
        if constexpr (C::DK == 0) { __syncthreads(); park_slices(sw); }
    }
    // ---- fold the K slices of conv 1's weight gradient (roles S and D) through LDS, fixed order, into this workgroup's
    //      partial slab of layer 0 ----
    // (96 x 96, where only role S holds slices: it parks them at the end of its own block -- park_slices below -- so that the
    //  accumulators are not live, i.e. not registers, in the other roles: at 14 waves a wave has 128)
    if constexpr (C::DK != 0) {
        __syncthreads();
        if (wave < C::ND || wave >= C::ND + 4) park_slices(wave >= C::ND + 4 ? wave - C::ND - 4 : 4 + wave);
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += C::NT) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < C::NSL; ++k) v += lds[k * 1024 + e];
        slab0[e] = v;
    }
    if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < C::NSL; ++k) v += lds[C::NSL * 1024 + k * 32 + tid];
        slab0[1024 + tid] = v;
    }
}

template <class C>
int launch_tail2(var_ctx* c, hipStream_t s, int B) {
    ProfScope prof(c, s, TAG_IMG_DGRAD0 + 1);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_tail2_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             C::LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const int G = B < kTail2G ? B : kTail2G;
    c->wg_groups[0] = G;
    c->wg_groups[1] = G;
    hipLaunchKernelGGL(img_tail2_kernel<C>, dim3(G), dim3(C::NT), C::LDS_BYTES, s, c->gact[2], c->wpack + c->kl.img_d[1],
                       c->act[1], c->saved_image, c->saved_bstride, c->saved_index, c->slabs + img_slab_offset(0),
                       c->slabs + img_slab_offset(1), B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
}  // namespace

// weight gradient of conv 2 + data gradient of conv 2 + weight gradient of conv 1 at 84 x 84: consumes gact[2], act[1] and the
// saved input image; leaves layer 0's and layer 1's slabs (c->wg_groups[0..1] of them) for launch_img_wgrad_reduce
int launch_img_bwd_tail2(var_ctx* c, hipStream_t s, int B) {
    if (c->H == 96) return c->saved_u8 ? launch_tail2<Tail2Cfg<true, 48, 4, false>>(c, s, B) : launch_tail2<Tail2Cfg<false, 48, 4, false>>(c, s, B);
    return c->saved_u8 ? launch_tail2<Tail2Cfg<true, 42, 6, true>>(c, s, B) : launch_tail2<Tail2Cfg<false, 42, 6, true>>(c, s, B);
}
