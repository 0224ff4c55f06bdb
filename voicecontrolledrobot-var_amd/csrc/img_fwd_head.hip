// Head of the image CNN forward in ONE kernel: conv 1 (3 -> 32, u8/f32 image) and conv 2 (32 -> 32),
// each Conv2d 3x3 stride 2 pad 1 + bias + ReLU (models/pretext/arm_pretext_model.py:9-12).
//
// The first activation map (B x 32 x H/2 x H/2 floats) is the largest tensor of the step.  Unfused it is
// written by the first conv kernel and read back (with a 7/6 halo) by the second.  Here a persistent
// workgroup computes, per tile (R2 output rows of conv 2, for NU images/bands):
//   1. the 2*R2+1 rows of act1 the tile needs, on the matrix cores, from the image band staged in LDS
//      (D[n][pixel], K = 27 (tap, c) pairs; the u8 -> f32 "/255" of dataset.py:67-68 is a 256-entry LDS table);
//      bias + ReLU; the tile goes to LDS (the B operand of conv 2), the rows the band owns also go to HBM
//      (backward needs act1)
//   2. conv 2 from that LDS tile: (pixel block) x (filter row ky) wave grid, filter resident in LDS,
//      the three ky partials folded through LDS in a fixed order; bias + ReLU; act2 to HBM.
// act1 is written once and not read again in the forward pass.
#include <stdlib.h>

#include <type_traits>

#include "img_stage.h"

template <int H1_, bool U8_, int R2_, int NU_>
struct HeadCfg {
    static constexpr int H1 = H1_, W1 = H1_, R2 = R2_, NU = NU_;   // act1 plane (42 / 48); NU bands per tile
    static constexpr bool U8 = U8_;
    static constexpr int CH = 32;
    static constexpr int HI = 2 * H1;                        // image (84 / 96)
    static constexpr int HO2 = H1 / 2, WO2 = HO2;            // act2 plane (21 / 24)
    static constexpr int NB = HO2 / R2;                      // bands per image
    static constexpr int IR1 = 2 * R2 + 1;                   // act1 rows per band (first one shared with the band above)
    static constexpr int IRI = 2 * IR1 + 1;                  // image rows per band
    static constexpr int PWI = HI + 2;                       // image band row: col = ix + 1
    static constexpr int PLANE_I = IRI * PWI, UNIT_I = 3 * PLANE_I;
    static constexpr int PW1 = W1 + 2;                       // act1 band row: col = x + 1
    static constexpr int PLANE_1 = IR1 * PW1, UNIT_1 = CH * PLANE_1;
    static constexpr int NPX1 = NU * IR1 * W1;               // act1 pixels per tile
    static constexpr int NP1 = (NPX1 + 31) / 32;             // ... in 32-pixel blocks
    static constexpr int NPX2 = NU * R2 * WO2;               // act2 pixels per tile
    static constexpr int NPB2 = (NPX2 + 31) / 32;
    static constexpr int NW = 3 * NPB2, NT = NW * 64;        // wave = ky * NPB2 + pixel block
    static constexpr int NR1 = (NP1 + NW - 1) / NW;          // conv-1 rounds
    static constexpr int KS1 = 14;                           // k-steps of conv 1 (27 -> 28)
    static constexpr int IMS = 0;
    static constexpr int A1S = (NU * UNIT_I + 3) & ~3;
    static constexpr int WDS = (A1S + NU * UNIT_1 + 3) & ~3; // conv-2 filter Wf[k][n], k = tap*CH + c
    static constexpr int LUT = WDS + 9 * CH * CH;            // x / 255.f
    static constexpr int BIA = LUT + 256;                    // bias of conv 1 (32) and conv 2 (32)
    static constexpr int LDS_FLOATS = BIA + 64;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(HO2 % R2 == 0, "whole bands");
    static_assert(2 * NPB2 * 1024 <= NU * UNIT_1, "the fold scratch aliases the act1 tile");
    static_assert(NU <= 2, "at most two bands per tile");
};

PH_DECL();
#ifdef VAR_PHASES
extern "C" int var_debug_phases_head(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

template <class C>
__global__ void __launch_bounds__(C::NT)
img_fwd_head_kernel(const void* __restrict__ image, long bstride, const int* __restrict__ bidx,
                    const float* __restrict__ wp1, const float* __restrict__ bias1,
                    const float* __restrict__ wp2, const float* __restrict__ bias2,
                    float* __restrict__ y1, float* __restrict__ y2, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NT;
    using XT = typename std::conditional<C::U8, uint8_t, float>::type;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int ntiles = (total_units + C::NU - 1) / C::NU;
    const float* lut = lds + C::LUT;

    PH_INIT(7);
    // ---- one-time: conv-2 filter and the u8 table into LDS, conv-1 filter into registers, pad columns ----
    for (int e = tid; e < 9 * C::CH * C::CH / 4; e += NT) ((float4*)(lds + C::WDS))[e] = ((const float4*)wp2)[e];
    if (tid < 256) lds[C::LUT + tid] = (float)tid / 255.f;
    if (tid < 32) { lds[C::BIA + tid] = bias1[tid]; lds[C::BIA + 32 + tid] = bias2[tid]; }
    lds_zero_cols<NT>(lds + C::IMS, C::NU * 3 * C::IRI, C::PWI, 0, 1, tid);
    lds_zero_cols<NT>(lds + C::IMS, C::NU * 3 * C::IRI, C::PWI, C::HI + 1, C::PWI - C::HI - 1, tid);
    float w1[C::KS1];
    int osel[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) {
        // k = 2s + half; tap = k / 3, c = k % 3 (k = 27 is the zero row of the packed filter)
        w1[s] = wp1[(2 * s + half) * C::CH + l31];
        const int k0 = 2 * s, k1 = 2 * s + 1 > 26 ? 26 : 2 * s + 1;
        const int o0 = (k0 % 3) * C::PLANE_I + ((k0 / 3) / 3) * C::PWI + ((k0 / 3) % 3);
        const int o1 = (k1 % 3) * C::PLANE_I + ((k1 / 3) / 3) * C::PWI + ((k1 / 3) % 3);
        osel[s] = half ? o1 : o0;
    }

    // ---- conv-1 lane constants, per round: this lane's act1 pixel ----
    int px_u[C::NR1], px_row[C::NR1], px_col[C::NR1];
    bool px_ok[C::NR1];
#pragma unroll
    for (int i = 0; i < C::NR1; ++i) {
        int p = (wave + i * C::NW) * 32 + l31;
        px_ok[i] = p < C::NPX1;
        if (!px_ok[i]) p = 0;
        px_u[i] = p / (C::IR1 * C::W1);
        const int q = p - px_u[i] * (C::IR1 * C::W1);
        px_row[i] = q / C::W1;
        px_col[i] = q - px_row[i] * C::W1;
    }
    // ---- conv-2 lane constants ----
    const int blk2 = wave % C::NPB2, ky = wave / C::NPB2;
    int p2 = blk2 * 32 + l31;
    const bool p2ok = p2 < C::NPX2;
    if (!p2ok) p2 = 0;
    const int u2 = p2 / (C::R2 * C::WO2), q2 = p2 - u2 * (C::R2 * C::WO2);
    const int oyl2 = q2 / C::WO2, ox2 = q2 - oyl2 * C::WO2;
    const int pixoff2 = C::A1S + u2 * C::UNIT_1 + (2 * oyl2 + ky) * C::PW1 + 2 * ox2 + half * C::PLANE_1;
    const int wl2 = C::WDS + ((ky * 3) * C::CH + half) * C::CH + l31;      // + (kx*CH + 2*c2)*CH

    // The next tile's image band is fetched and stored by the waves that idle during the fold (ky > 0): their loads
    // fly during conv 2, their LDS stores happen while the ky = 0 waves run the epilogue.
    constexpr int CPW = C::NW >= 9 ? 6 : 3;                      // copy waves (lanes must split evenly over the 3 * NU planes)
    constexpr int CPT = CPW * 64;
    const bool copier = wave >= C::NPB2 && wave < C::NPB2 + CPW;
    const int ctid = tid - C::NPB2 * 64;
    BandCopy<3, C::HI, C::HI, C::IRI, C::U8, CPT, C::NU> cx;
    struct Tile { const XT* im0; const XT* im1; int rx0, rx1; bool ok0, ok1; };
    auto gather_rows = [&](int tile, int& gi0, int& gi1) {       // dataset rows (optional gather), one tile ahead
        const int u0 = tile * C::NU, u1 = u0 + C::NU - 1;
        const int b0 = u0 < total_units ? u0 / C::NB : 0, b1 = u1 < total_units ? u1 / C::NB : 0;
        gi0 = bidx ? bidx[b0] : b0;
        gi1 = bidx ? bidx[b1] : b1;
    };
    auto make_tile = [&](int tile, int gi0, int gi1) {
        Tile t;
        const int u0 = tile * C::NU, u1 = u0 + C::NU - 1;
        t.ok0 = u0 < total_units; t.ok1 = u1 < total_units;
        const int band0 = t.ok0 ? u0 % C::NB : 0, band1 = t.ok1 ? u1 % C::NB : 0;
        t.im0 = (const XT*)image + (size_t)gi0 * bstride;
        t.im1 = (const XT*)image + (size_t)gi1 * bstride;
        t.rx0 = 4 * band0 * C::R2 - 3; t.rx1 = 4 * band1 * C::R2 - 3;      // image row of band row 0
        return t;
    };

    PH(0);
    int tile = blockIdx.x;
    int gi0 = 0, gi1 = 0;
    if (tile < ntiles) {
        gather_rows(tile, gi0, gi1);
        const Tile t = make_tile(tile, gi0, gi1);
        if (copier) cx.issue(t.im0, t.im1, t.rx0, t.rx1, ctid);
        __syncthreads();                                   // table is in place
        if (copier) cx.template store<C::UNIT_I, C::PLANE_I, C::PWI, 1>(lds + C::IMS, lut, t.rx0, t.rx1, t.ok0, t.ok1, ctid);
    }
    if (tile + (int)gridDim.x < ntiles) gather_rows(tile + gridDim.x, gi0, gi1);
    PH(1);
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                   // (1) image band staged; previous fold is done with LDS
        PH(2);
        const int next = tile + gridDim.x;
        const bool more = next < ntiles;
        int tid_t = tid, ctid_t = ctid;                    // opaque copies: see BandCopy
        asm volatile("" : "+v"(tid_t));
        asm volatile("" : "+v"(ctid_t));
        const Tile tn = make_tile(more ? next : tile, gi0, gi1);
        if (next + (int)gridDim.x < ntiles) gather_rows(next + gridDim.x, gi0, gi1);
        if (more && copier) cx.issue(tn.im0, tn.im1, tn.rx0, tn.rx1, ctid_t);     // in flight during conv 1 and conv 2
        // pad columns of the act1 tile (the fold scratch of the previous tile ran over them)
        lds_zero_cols<NT>(lds + C::A1S, C::NU * C::CH * C::IR1, C::PW1, 0, 1, tid_t);
        lds_zero_cols<NT>(lds + C::A1S, C::NU * C::CH * C::IR1, C::PW1, C::W1 + 1, C::PW1 - C::W1 - 1, tid_t);

        // ---- conv 1: act1 rows of the tile ----
#pragma unroll
        for (int i = 0; i < C::NR1; ++i) {
            if ((wave + i * C::NW) * 32 >= C::NPX1) continue;        // wave-uniform: no block in this round
            const int pixoff = C::IMS + px_u[i] * C::UNIT_I + (2 * px_row[i]) * C::PWI + 2 * px_col[i];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < C::KS1; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[s], lds[pixoff + osel[s]], acc, 0, 0, 0);
            // epilogue: bias + ReLU; tile -> LDS, owned rows -> HBM
            const int unit = tile * C::NU + px_u[i];
            const bool uok = unit < total_units;
            const int b = uok ? unit / C::NB : 0, band = uok ? unit - b * C::NB : 0;
            const int r1 = 2 * band * C::R2 - 1 + px_row[i];         // act1 row (-1: conv 2's zero padding)
            const bool live = px_ok[i] && uok && r1 >= 0;
            const bool own = live && px_row[i] >= 1;
            const int lo = C::A1S + px_u[i] * C::UNIT_1 + px_row[i] * C::PW1 + 1 + px_col[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[r] + lds[C::BIA + n];
                v = (live && v > 0.f) ? v : 0.f;
                acc[r] = v;
            }
            if (px_ok[i]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[lo + ((r & 3) + 8 * (r >> 2) + 4 * half) * C::PLANE_1] = acc[r];
            }
            if (own) {
                float* gp = y1 + (size_t)b * C::CH * C::H1 * C::W1 + r1 * C::W1 + px_col[i] + 4 * half * C::H1 * C::W1;
#pragma unroll
                for (int r = 0; r < 16; ++r) gp[((r & 3) + 8 * (r >> 2)) * C::H1 * C::W1] = acc[r];
            }
        }
        PH(3);
        __syncthreads();                                   // (2) act1 tile complete; image band is dead
        PH(4);

        // ---- conv 2: this wave's filter row of its pixel block ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            constexpr int UC = 4, NCH = (C::CH / 2) / UC;        // chunks of UC channel pairs x 3 kx
            float bb[2][UC][3], wa[2][UC][3];
            auto fetch = [&](int buf, int ch) {
#pragma unroll
                for (int u = 0; u < UC; ++u)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        bb[buf][u][kx] = lds[pixoff2 + 2 * (ch * UC + u) * C::PLANE_1 + kx];
                        wa[buf][u][kx] = lds[wl2 + (kx * C::CH + 2 * (ch * UC + u)) * C::CH];
                    }
            };
            fetch(0, 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) fetch((ch + 1) & 1, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UC; ++u)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ch & 1][u][kx], bb[ch & 1][u][kx], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PH(5);
        // ---- fold the filter rows (fixed order: ky 0 + ky 1 + ky 2) ----
        __syncthreads();                                   // (3) the act1 tile is dead
        PH(6);
        if (ky > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[C::A1S + ((ky - 1) * C::NPB2 + blk2) * 1024 + r * 64 + lane] = acc[r];
        }
        __syncthreads();                                   // (4)
        PH(7);
        if (more && copier) cx.template store<C::UNIT_I, C::PLANE_I, C::PWI, 1>(lds + C::IMS, lut, tn.rx0, tn.rx1, tn.ok0, tn.ok1, ctid_t);
        if (ky == 0) {
            const int unit = tile * C::NU + u2;
            const bool uok = unit < total_units;
            const int b = uok ? unit / C::NB : 0, band = uok ? unit - b * C::NB : 0;
            float* yp = y2 + (size_t)b * C::CH * C::HO2 * C::WO2 + band * C::R2 * C::WO2 + q2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[r] + lds[C::A1S + blk2 * 1024 + r * 64 + lane];
                v += lds[C::A1S + (C::NPB2 + blk2) * 1024 + r * 64 + lane];
                v += lds[C::BIA + 32 + n];
                if (uok && p2ok) yp[(size_t)n * C::HO2 * C::WO2] = v > 0.f ? v : 0.f;
            }
        }
        PH(8);
    }
}

//                    H1   U8   R2 NU
// (measured, round 2: one band per tile with each wave's conv-2 filter row in registers -- 56 KB, 6 waves, two workgroups
//  per CU -- runs at 61 us instead of 52, and starting the second workgroup of each CU later only adds the delay: the
//  per-tile fixed costs double with half-size tiles; the 12-wave form with the filter rows in registers spills, 57 us)
using H84u = HeadCfg<42, true, 3, 2>;      // 2 bands: 4 pixel blocks x 3 filter rows = 12 waves, 3 per SIMD
using H84f = HeadCfg<42, false, 3, 2>;
using H96u = HeadCfg<48, true, 4, 1>;      // 1 band: 3 pixel blocks x 3 filter rows = 9 waves
using H96f = HeadCfg<48, false, 4, 1>;

template <class C>
static int launch_head(var_ctx* c, hipStream_t s, const void* image, long bstride, const int* bidx,
                       const float* params, int B) {
    ProfScope prof(c, s, TAG_IMG_FWD0 + 1);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_fwd_head_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const int ntiles = (B * C::NB + C::NU - 1) / C::NU;
    const int G = ntiles < 256 ? ntiles : 256;
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    hipLaunchKernelGGL(img_fwd_head_kernel<C>, dim3(G), dim3(C::NT), C::LDS_BYTES, s, image, bstride, bidx,
                       c->wpack + K.img_f[0], params + L.img_b[0], c->wpack + K.img_f[1], params + L.img_b[1],
                       c->act[1], c->act[2], B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// conv 1 + conv 2 of the image CNN; leaves act[1] and act[2]
int launch_img_fwd_head(var_ctx* c, hipStream_t s, const float* params, const void* image, int is_u8, long bstride,
                        const int* image_index, int B) {
    if (c->H == 84) return is_u8 ? launch_head<H84u>(c, s, image, bstride, image_index, params, B)
                                 : launch_head<H84f>(c, s, image, bstride, image_index, params, B);
    return is_u8 ? launch_head<H96u>(c, s, image, bstride, image_index, params, B)
                 : launch_head<H96f>(c, s, image, bstride, image_index, params, B);
}
