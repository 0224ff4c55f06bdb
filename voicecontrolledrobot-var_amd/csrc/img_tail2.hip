// End of the image CNN backward at 84 x 84 in ONE role-specialised kernel: the weight gradient of conv 2, the data
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

template <bool U8_>
struct Tail2Cfg {
    static constexpr bool U8 = U8_;
    static constexpr int CH = 32, H1 = 42, W1 = 42, HI = 84, HO2 = 21, NB = 7;
    static constexpr int RB = 6;                              // gact1 rows per band
    // act1 band: EXACTLY img_head2.hip's tile, in LDS and in HBM ([image][band][channel][A1_PLANE]): per channel 2 pad floats,
    // then 7 rows of 42 with no padding between them (row 0 = the halo row above the band).  The cell x = -1 of a row is the
    // previous row's last cell: role W zeroes that operand (pixels ox = 0, taps kx = 0) instead of reading a pad.
    static constexpr int A1_ROWS = 7, A1_PITCH = W1, A1_ROW0 = 2, A1_PLANE = 300, A1_FLOATS = CH * A1_PLANE;
    static constexpr int G2_ROWS = 4, G2_PITCH = 22, G2_PLANE = 112, G2_FLOATS = CH * G2_PLANE;                   // col = ox, col 21 = 0
    static constexpr int G1_PLANE = RB * W1, G1_FLOATS = CH * G1_PLANE;                                           // [c][6 x 42]
    static constexpr int IM_ROWS = 2 * RB + 1, IM_PITCH = HI + 4, IM_PLANE = IM_ROWS * IM_PITCH, IM_FLOATS = 3 * IM_PLANE;  // col = x + 4
    static constexpr int A1S = 0, G2S = A1S + 2 * A1_FLOATS, G1S = G2S + 2 * G2_FLOATS, IMS = G1S + G1_FLOATS;
    static constexpr int LUT = IMS + IM_FLOATS, ZCELL = LUT + 256;
    static constexpr int LDS_FLOATS = ZCELL + 4;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int NT = 768;                            // 4 waves of each role
    static constexpr int NPX2 = 3 * HO2;                      // conv-2 output pixels a band's weight gradient sums over (63)
    static constexpr int KS_W = (NPX2 + 3) / 4;               // its k-steps (16)
    static constexpr int NPX1 = RB * W1;                      // gact1 pixels of a band (252)
    static constexpr int KS_1 = NPX1 / 4;                     // k-steps of conv 1's weight gradient (63)
    static constexpr int N_A1 = A1_FLOATS / 4;                 // float4 of an act1 band (one contiguous block)
    static constexpr int N_G2 = CH * G2_ROWS * HO2;           // floats of a gact2 band
    static constexpr int N_IM = 3 * IM_ROWS * (HI / 4);       // 4-pixel groups of an image band
    static constexpr int L_A1 = (N_A1 + 255) / 256,   /* 1 KiB chunks per wave of role S */ L_G2 = (N_G2 + 255) / 256, L_IM = (N_IM + 255) / 256;
    static constexpr int SLAB0 = 32 * 32 + 32, SLAB1 = 32 * 9 * 32 + 32;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(8 * 1024 + 256 <= 2 * A1_FLOATS, "the end-of-kernel fold scratch aliases the act1 buffers");
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
    for (int e = tid; e < 3 * C::IM_ROWS; e += C::NT) lds[C::IMS + e * C::IM_PITCH + 3] = 0.f;                     // x = -1
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
    auto wgrad1_chunk = [&](auto ck, int s_lo, int s_hi) {
        int m = 4 * s_lo + q;
        int yl = m / C::W1, x = m - yl * C::W1;
        const int ga = C::G1S + l15 * C::G1_PLANE;
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
            b0[u] = lds[cbase[0] + po];
            b1[u] = lds[cbase[1] + po];
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

    if (wave < 4) {
        // =====================================================================================================
        // role D: data gradient of conv 2 for this wave's two pixel tiles and channel half, masked, into LDS
        // =====================================================================================================
        const int ct = wave & 1, pp = wave >> 1;
        float A[72];                                           // A[i = c][k = n] of (tap, k-step s): Wd[tap][4 s + q][16 ct + l15]
#pragma unroll
        for (int t = 0; t < 72; ++t) A[t] = wd[((t >> 3) * 32 + 4 * (t & 7) + q) * 32 + 16 * ct + l15];
#pragma unroll
        for (int t = 0; t < 72; ++t) asm volatile("" : "+v"(A[t]));      // waited for here, not inside the band loop (img_head2.hip)
        int mj[2], mi[2];
        bool mok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int m = (2 * pp + u) * 16 + l15;                   // pixel of the parity class: (row pair j, column pair i)
            mok[u] = m < C::NPX2;
            if (!mok[u]) m = 0;
            mj[u] = m / C::HO2; mi[u] = m - mj[u] * C::HO2;
        }
        __syncthreads();                                       // (P) band 0 staged
        PHR(0);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();                                   // (A)
            PHR(1);
            const int g2 = C::G2S + (j & 1) * C::G2_FLOATS + q * C::G2_PLANE;
            const int a1 = C::A1S + (j & 1) * C::A1_FLOATS;
            // nine (parity class, tap) chunks of 8 k-steps x 2 pixel tiles; the operands of chunk k + 1 are read before the 16
            // MFMAs of chunk k (pinned with scheduling barriers: left alone hipcc sinks every read to its use and each pair of
            // MFMAs then waits out an LDS latency); a class's accumulators are masked and stored behind its last chunk
            constexpr int kCls[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};                   // class = 2 py + px
            constexpr int kKy[9] = {1, 1, 1, 0, 2, 0, 0, 2, 2}, kKx[9] = {1, 0, 2, 1, 1, 0, 2, 0, 2};
            // (opaque per band: hipcc otherwise hoists the sixteen mask / store addresses of the epilogues out of the band loop)
            int mjb[2] = {mj[0], mj[1]}, mib[2] = {mi[0], mi[1]};
#pragma unroll
            for (int u = 0; u < 2; ++u) { asm volatile("" : "+v"(mjb[u])); asm volatile("" : "+v"(mib[u])); }
            const int lb0 = g2 + mjb[0] * C::G2_PITCH + mib[0], lb1 = g2 + mjb[1] * C::G2_PITCH + mib[1];
            // (operands in half chunks of 4 k-steps: 16 registers in flight beside the filter's 72)
            float bb[2][2][4];
            auto fetch = [&](int buf, int h) {                 // half chunk h = 2 k + (upper half)
                const int k = h >> 1, s0 = 4 * (h & 1);
                // gy row j + 1 for ky = 0 (odd rows), column i + 1 for kx = 0 (odd columns)
                const int off = (kKy[k] == 0 ? C::G2_PITCH : 0) + (kKx[k] == 0 ? 1 : 0);
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    bb[buf][0][ss] = lds[lb0 + off + 4 * (s0 + ss) * C::G2_PLANE];
                    bb[buf][1][ss] = lds[lb1 + off + 4 * (s0 + ss) * C::G2_PLANE];
                }
            };
            f32x4t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            fetch(0, 0);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                const int k = h >> 1;
                if (h + 1 < 18) fetch((h + 1) & 1, h + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int tap = kKy[k] * 3 + kKx[k];
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tap * 8 + 4 * (h & 1) + ss], bb[h & 1][0][ss], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tap * 8 + 4 * (h & 1) + ss], bb[h & 1][1][ss], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(h & 1)) continue;
                if (k + 1 == 9 || kCls[k + 1] != kCls[k]) {
                    // ReLU mask of act1 at (row 2 j + py, column 2 i + px), then the band of gact1 in LDS
                    const int py = kCls[k] >> 1, px = kCls[k] & 1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int yl = 2 * mjb[u] + py, x = 2 * mib[u] + px;
                        const int mo = a1 + (16 * ct + 4 * q) * C::A1_PLANE + C::A1_ROW0 + (yl + 1) * C::A1_PITCH + x;
                        const int go = C::G1S + (16 * ct + 4 * q) * C::G1_PLANE + yl * C::W1 + x;
                        float mv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) mv[r] = lds[mo + r * C::A1_PLANE];
                        if (mok[u]) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) lds[go + r * C::G1_PLANE] = mv[r] > 0.f ? acc[u][r] : 0.f;
                        }
                        acc[u] = {0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            PHR(2);
            __syncthreads();                                   // (B) the band of gact1 is complete
            PHR(3);
            // conv 1's weight gradient: this wave's 8 k-steps, in two chunks of 4 (the filter's 72 registers leave no room for 8)
            wgrad1_chunk(std::integral_constant<int, 4>{}, 32 + 8 * wave, 32 + 8 * wave + 4);
            wgrad1_chunk(std::integral_constant<int, 4>{}, 32 + 8 * wave + 4, 32 + 8 * wave + 8 < C::KS_1 ? 32 + 8 * wave + 8 : C::KS_1);
            PHR(4);
        }
        PHR_FLUSH();
    } else if (wave < 8) {
        // =====================================================================================================
        // role W: weight gradient of conv 2, the nine taps of this wave's (16 n, 16 c) block, summed over all bands
        // =====================================================================================================
        const int w = wave - 4, nt = w & 1, ctl = w >> 1;
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
                if (ss + 1 == C::KS_W / 2) { PHR(2); __syncthreads(); PHR(3); }                 // (B): reads only, either side of it
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
    } else {
        // =====================================================================================================
        // role S: staging of the next band + weight gradient of conv 1 over the band of gact1 in LDS
        // =====================================================================================================
        const int sw = wave - 8, stid0 = tid - 512;
        float r_g2[C::L_G2];
        uint32_t r_u8[C::U8 ? C::L_IM : 1];
        float4 r_f4[C::U8 ? 1 : C::L_IM];
        // per-lane element tables, computed once (they do not depend on the band): gact2 element e -> (global offset inside the
        // image's band rows | LDS offset | row r) packed in one register; image 4-pixel group e -> (global | LDS | row) likewise
        uint32_t t_g2[C::L_G2], t_im[C::L_IM];
#pragma unroll
        for (int i = 0; i < C::L_G2; ++i) {
            const int e = stid0 + 256 * i;
            const int ee = e < C::N_G2 ? e : 0;
            const int ch = ee / (C::G2_ROWS * C::HO2), rem = ee - ch * (C::G2_ROWS * C::HO2);
            const int r = rem / C::HO2, ox = rem - r * C::HO2;
            // global: ch * 441 + r * 21 + ox (< 2^14); LDS: ch * G2_PLANE + r * G2_PITCH + ox (< 2^12); row r; bit 31 = no element
            t_g2[i] = (uint32_t)(ch * C::HO2 * C::HO2 + r * C::HO2 + ox) | ((uint32_t)(ch * C::G2_PLANE + r * C::G2_PITCH + ox) << 14) |
                      ((uint32_t)r << 26) | (e < C::N_G2 ? 0u : 0x80000000u);
        }
#pragma unroll
        for (int i = 0; i < C::L_IM; ++i) {
            const int e = stid0 + 256 * i;
            const int ee = e < C::N_IM ? e : 0;
            const int c = ee / (C::IM_ROWS * (C::HI / 4)), rem = ee - c * (C::IM_ROWS * (C::HI / 4));
            const int r = rem / (C::HI / 4), g = rem - r * (C::HI / 4);
            // global: c * 84 * 84 + r * 84 + 4 g (< 2^15); LDS: c * IM_PLANE + r * IM_PITCH + 4 + 4 g (< 2^12); row r (< 16)
            t_im[i] = (uint32_t)(c * C::HI * C::HI + r * C::HI + 4 * g) | ((uint32_t)(c * C::IM_PLANE + r * C::IM_PITCH + 4 + 4 * g) << 15) |
                      ((uint32_t)r << 27) | (e < C::N_IM ? 0u : 0x80000000u);
        }
        auto issue_a1g2 = [&](int j, int stid) {
            const int band = j % C::NB, b = tile_img(j);
            // the act1 band: ONE contiguous block in HBM in the tile's own layout (img_head2.hip) -> straight into LDS by
            // LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes land at the wave's base + 16 l), 1 KiB per wave instruction,
            // no registers, no store pass; it lands during this band's matrix work and is waited for before the next barrier (A)
            const float* pa = act1 + (size_t)b * kAct1TiledFloats + (size_t)band * C::A1_FLOATS;
            float* da = lds + C::A1S + (j & 1) * C::A1_FLOATS;
#pragma unroll
            for (int i = 0; i < C::L_A1; ++i) {
                const int k = sw + 4 * i;                                              // 1 KiB chunk of this wave
                if (k * 64 + lane < C::N_A1)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + k * 256 + 4 * lane),
                                                     (__attribute__((address_space(3))) void*)(da + k * 256), 16, 0, 0);
            }
            const float* pg = gy + (size_t)b * C::CH * C::HO2 * C::HO2 + 3 * band * C::HO2;
#pragma unroll
            for (int i = 0; i < C::L_G2; ++i) {
                const int r = (t_g2[i] >> 26) & 3;
                const int back = (3 * band + r >= C::HO2) ? C::HO2 : 0;               // a row below the image: read the row above (discarded)
                r_g2[i] = pg[(int)(t_g2[i] & 0x3fff) - back];
            }
        };
        auto store_a1g2 = [&](int j, int stid) {
            const int band = j % C::NB;
            float* dg = lds + C::G2S + (j & 1) * C::G2_FLOATS;
#pragma unroll
            for (int i = 0; i < C::L_G2; ++i) {
                const int r = (t_g2[i] >> 26) & 3;
                if (!(t_g2[i] >> 31)) dg[(t_g2[i] >> 14) & 0xfff] = (3 * band + r < C::HO2) ? r_g2[i] : 0.f;
            }
        };
        auto issue_img = [&](int j) {
            const int band = j % C::NB, b = tile_img(j);
            const int gi = bidx ? bidx[b] : b;
            const int iy0 = 2 * C::RB * band - 1;
            const XT* im = (const XT*)image + (size_t)gi * bstride;
#pragma unroll
            for (int i = 0; i < C::L_IM; ++i) {
                const int r = (t_im[i] >> 27) & 15;
                const int iy = iy0 + r;
                const int fix = iy < 0 ? C::HI : 0;                                   // row -1: read row 0 (discarded)
                const XT* src = im + (int)(t_im[i] & 0x7fff) + iy0 * C::HI + fix;
                if constexpr (C::U8) r_u8[i] = *(const uint32_t*)src;
                else r_f4[i] = *(const float4*)src;
            }
        };
        auto store_img = [&](int j) {
            const int band = j % C::NB;
            const int iy0 = 2 * C::RB * band - 1;
#pragma unroll
            for (int i = 0; i < C::L_IM; ++i) {
                if (t_im[i] >> 31) continue;
                const int r = (t_im[i] >> 27) & 15;
                const bool rok = iy0 + r >= 0;                                        // (rows >= HI do not occur: 12 * 6 + 11 = 83)
                float4 v;
                if constexpr (C::U8) {
                    const uint32_t wv = rok ? r_u8[i] : 0u;             // lut[0] = 0
                    v = make_float4(lut[wv & 0xff], lut[(wv >> 8) & 0xff], lut[(wv >> 16) & 0xff], lut[wv >> 24]);
                } else {
                    v = rok ? r_f4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *(float4*)(lds + C::IMS + ((t_im[i] >> 15) & 0xfff)) = v;
            }
        };
        if (ntl > 0) { issue_a1g2(0, stid0); store_a1g2(0, stid0); issue_img(0); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // (P)
        PHR(0);
        for (int j = 0; j < ntl; ++j) {
            __syncthreads();                                   // (A)
            PHR(1);
            // (an opaque copy of the lane id per band: hipcc otherwise hoists every element's index math out of the band
            //  loop and keeps ~150 registers of it alive -- or in scratch -- for the life of the kernel)
            int stid = stid0;
            asm volatile("" : "+v"(stid));
            store_img(j);
            if (j + 1 < ntl) { issue_img(j + 1); issue_a1g2(j + 1, stid); }   // a whole band of matrix work ahead of their use
            PHR(2);
            __syncthreads();                                   // (B) gact1 band complete (role D), image band stored (this role)
            PHR(3);
            wgrad1_chunk(std::integral_constant<int, 8>{}, 8 * sw, 8 * sw + 8);   // k-steps [0, 32) here, [32, 63) in role D
            PHR(4);
            asm volatile("" : "+v"(stid));
            if (j + 1 < ntl) store_a1g2(j + 1, stid);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the band's LDS-DMA has landed (and, with it, the image loads)
            PHR(5);
        }
        PHR_FLUSH();
    }
    // ---- fold the eight K slices of conv 1's weight gradient (roles S and D) through LDS, fixed order, into this workgroup's
    //      partial slab of layer 0 ----
    __syncthreads();
    if (wave < 4 || wave >= 8) {
        const int sl = wave >= 8 ? wave - 8 : 4 + wave;        // slice = its range of k-steps, ascending
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
        if (q == 0) { lds[8192 + sl * 32 + l15] = b0; lds[8192 + sl * 32 + 16 + l15] = b1; }
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += C::NT) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += lds[k * 1024 + e];
        slab0[e] = v;
    }
    if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += lds[8192 + k * 32 + tid];
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
    return c->saved_u8 ? launch_tail2<Tail2Cfg<true>>(c, s, B) : launch_tail2<Tail2Cfg<false>>(c, s, B);
}
