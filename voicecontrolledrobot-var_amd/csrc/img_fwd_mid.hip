// Middle and end of the image CNN forward in ONE kernel: conv 3, conv 4 and conv 5 (32 -> 64 -> 64 -> 64
// channels, each Conv2d 3x3 stride 2 pad 1 + bias + ReLU, models/pretext/arm_pretext_model.py:13-18).
//
// From the third conv on an image's activations are small (56 KB -> 31 KB -> 9 KB -> 2 KB): one workgroup owns one
// image and walks the three layers with the activations resident in LDS.  Per layer: implicit GEMM on the
// matrix cores exactly as img_conv_fwd.hip (D[n][pixel], filter rows prefetched from the packed image in L2,
// B operand from the LDS tile), 12 waves = item groups x filter-row / channel slices folded through LDS in a
// fixed order; the epilogue (bias + ReLU) writes each activation to HBM (backward needs it) AND into the
// zero-padded LDS tile the next layer reads.  Two kernel launches, two activation round trips through HBM and
// two pipeline fill/drain phases per step disappear.
#include "img_stage.h"

namespace {
constexpr int MID_NW = 12, MID_NT = MID_NW * 64;

// LDS tile of a layer input with HIN x HIN planes: [c][2*HO+1 rows][2*HO+2 cols], cell (r, col) = input (r-1, col-1)
template <int HIN>
struct MidTile {
    static constexpr int HO = (HIN - 1) / 2 + 1;
    static constexpr int IR = 2 * HO + 1, PW = 2 * HO + 2, PLANE = IR * PW;
};

// One conv layer of the workgroup's image.  xs: input tile (LDS), red: fold scratch (LDS, may alias xs),
// xo: next layer's input tile (LDS, pre-zeroed pads) or nullptr, y: this image's output planes in HBM.
// NPB_: pixel blocks to lay the work out on (0 = just enough for the map); a padded count lets 144-pixel maps
// (96 x 96 inputs) share the 12-wave plan of the 121-pixel ones, the surplus lanes compute on pixel 0 and store nothing
template <int CIN, int COUT, int HIN, int NWI, int KY, int KC, int PLANE_O, int PW_O, int NPB_ = 0>
__device__ __forceinline__ void mid_layer(const float* __restrict__ xs, float* __restrict__ red, float* __restrict__ xo,
                                          const float* __restrict__ wp, const float* __restrict__ bias,
                                          float* __restrict__ y, int tid) {
    using T = MidTile<HIN>;
    constexpr int HO = T::HO, WO = HO, NPIX = HO * WO;
    constexpr int NPB = NPB_ ? NPB_ : (NPIX + 31) / 32, NBLK = COUT / 32, ITEMS = NPB * NBLK, IPW = ITEMS / NWI;
    static_assert(NPB * 32 >= NPIX, "the pixel blocks must cover the map");
    constexpr int KS = KY * KC;
    static_assert(NWI * KS == MID_NW && ITEMS % NWI == 0, "12 waves = item groups x K slices, whole items per wave");
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wv = wave % NWI, ks = wave / NWI;
    const int kyi = ks % KY, kci = ks / KY;
    constexpr int NKY = 3 / KY, CC = CIN / KC, SPT = CC / 2, U = SPT > 16 ? 16 : SPT, BPT = SPT / U, NBK = NKY * 3 * BPT;

    // a wave's IPW items share the channel block (nb) and differ in the pixel block: ONE filter stream per wave
    static_assert(NPB % IPW == 0, "a wave's items must share their channel block");
    constexpr int GPB = NPB / IPW;                           // pixel-block groups
    const int nb = wv / GPB, pb0 = (wv % GPB) * IPW;
    int pixoff[IPW];
    f32x16 acc[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        int p = (pb0 + i) * 32 + l31;
        if (p >= NPIX) p = 0;
        const int oy = p / WO, ox = p - oy * WO;
        pixoff[i] = (2 * oy) * T::PW + 2 * ox + half * T::PLANE + kyi * NKY * T::PW + kci * CC * T::PLANE;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    const float* wl = wp + nb * 32 + l31 + (half + (kyi * NKY * 3) * CIN + kci * CC) * COUT;
    float wbuf[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u) wbuf[0][u] = wl[(2 * u) * COUT];
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk) {
        const int tap = blk / BPT, c2b = (blk % BPT) * U;
        const int toff = (tap / 3) * T::PW + (tap % 3);
        if (blk + 1 < NBK) {
            const int ntap = (blk + 1) / BPT, nc2b = ((blk + 1) % BPT) * U;
#pragma unroll
            for (int u = 0; u < U; ++u) wbuf[(blk + 1) & 1][u] = wl[(ntap * CIN + 2 * (nc2b + u)) * COUT];
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the prefetch above this block's MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < IPW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wbuf[blk & 1][u], xs[pixoff[i] + 2 * (c2b + u) * T::PLANE + toff],
                                                              acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // Fold the K slices through LDS with the work spread over ALL waves: slice k owns accumulator registers
    // [16k/KS, 16(k+1)/KS) of its item group -- every wave parks the registers it does not own, then sums its own
    // ones over the other slices (fixed order) and runs the epilogue (bias + ReLU -> HBM and the next layer's LDS
    // tile) for them.  The input tile is dead after the first barrier (the scratch may alias it).
    auto own_lo = [](int k) { return (16 * k) / KS; };
    constexpr int SLOT = (KS - 1) * NWI * IPW * 64;          // floats per owned register index
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int ko = 0;                                          // owner of register r (compile-time after unrolling)
#pragma unroll
        for (int k = 1; k < KS; ++k) if (r >= (16 * k) / KS) ko = k;
        if (ko != ks) {
            const int sp = ks - (ks > ko ? 1 : 0);           // this wave's index among the owner's sources
#pragma unroll
            for (int i = 0; i < IPW; ++i)
                red[SLOT * r + ((sp * NWI + wv) * IPW + i) * 64 + lane] = acc[i][r];
        }
    }
    __syncthreads();
    (void)own_lo;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int p = (pb0 + i) * 32 + l31;
        const int oy = p / WO, ox = p - oy * WO;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int ko = 0;
#pragma unroll
            for (int k = 1; k < KS; ++k) if (r >= (16 * k) / KS) ko = k;
            if (ko != ks) continue;                          // wave-uniform
            float v = acc[i][r];
#pragma unroll
            for (int sp = 0; sp < KS - 1; ++sp) v += red[SLOT * r + ((sp * NWI + wv) * IPW + i) * 64 + lane];
            const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            v += bias[n];
            v = v > 0.f ? v : 0.f;
            if (p < NPIX) {
                y[n * NPIX + p] = v;
                if (xo) xo[n * PLANE_O + (oy + 1) * PW_O + ox + 1] = v;
            }
        }
    }
}

// The same layer on 16 x 16 x 4 matrix tiles (v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k][j = l & 15],
// D col = l & 15, row = 4 (l >> 4) + r): the late layers have 36 and 9 output pixels per image, which fill 56 % and
// 28 % of 32-pixel tiles but 75 % and 56 % of 16-pixel ones -- the matrix time of these layers drops by 25 % / 50 %.
// 12 waves = (pixel tile, 16-channel tile) items x KY filter-row slices; with KY = 1 there is no fold at all.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int CIN, int COUT, int HIN, int NWI, int KY, int PLANE_O, int PW_O>
__device__ __forceinline__ void mid_layer16(const float* __restrict__ xs, float* __restrict__ red, float* __restrict__ xo,
                                            const float* __restrict__ wp, const float* __restrict__ bias,
                                            float* __restrict__ y, int tid, float* __restrict__ flat = nullptr) {
    using T = MidTile<HIN>;
    constexpr int HO = T::HO, WO = HO, NPIX = HO * WO;
    constexpr int NPT = (NPIX + 15) / 16, NNT = COUT / 16;
    static_assert(NPT * NNT == NWI && NWI * KY == MID_NW, "12 waves = items x filter-row slices");
    const int lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l15 = lane & 15;
    const int wv = wave % NWI, ky0 = (wave / NWI) * (3 / KY);
    const int pt = wv % NPT, nt = wv / NPT;
    int p = pt * 16 + l15;
    const bool pok = p < NPIX;
    if (!pok) p = 0;
    const int oy = p / WO, ox = p - oy * WO;
    const int pixoff = (2 * oy + ky0) * T::PW + 2 * ox + kq * T::PLANE;
    const float* wl = wp + nt * 16 + l15 + (kq + ky0 * 3 * CIN) * COUT;
    constexpr int SPT = CIN / 4, U = 16, BPT = SPT / U, NBK = (3 / KY) * 3 * BPT;       // k-steps of 4 channels
    static_assert(SPT % U == 0, "whole prefetch blocks per tap");
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    float wbuf[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u) wbuf[0][u] = wl[(4 * u) * COUT];
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk) {
        const int tap = blk / BPT, c4b = (blk % BPT) * U;
        const int toff = (tap / 3) * T::PW + (tap % 3);
        if (blk + 1 < NBK) {
            const int ntap = (blk + 1) / BPT, nc4b = ((blk + 1) % BPT) * U;
#pragma unroll
            for (int u = 0; u < U; ++u) wbuf[(blk + 1) & 1][u] = wl[(ntap * CIN + 4 * (nc4b + u)) * COUT];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wbuf[blk & 1][u], xs[pixoff + 4 * (c4b + u) * T::PLANE + toff], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int ks = wave / NWI;
    if constexpr (KY > 1) {
        // fold the filter-row slices (fixed order) through LDS: 4 registers per lane, slice 0 finishes
        __syncthreads();
        if (ks > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((ks - 1) * NWI + wv) * 4 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int q = 1; q < KY; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += red[(((q - 1) * NWI + wv) * 4 + r) * 64 + lane];
        }
    }
    if (pok && ks == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nt * 16 + 4 * kq + r;
            float v = acc[r] + bias[n];
            v = v > 0.f ? v : 0.f;
            y[n * NPIX + p] = v;
            if (xo) xo[n * PLANE_O + (oy + 1) * PW_O + ox + 1] = v;
            if (flat) flat[n * NPIX + p] = v;                  // NCHW flatten order (the heads' input)
        }
    }
}

template <int H2>
struct MidCfg {
    using T2 = MidTile<H2>;
    static constexpr int H3 = T2::HO;
    using T3 = MidTile<H3>;
    static constexpr int H4 = T3::HO;
    using T4 = MidTile<H4>;
    static constexpr int H5 = T4::HO;
    // region A: conv-3 input tile, later the conv-3 fold scratch, then the conv-5 input tile (act4)
    // region B: conv-4 input tile (act3), later the fold scratch of conv 4 and conv 5
    static constexpr int X2 = 32 * T2::PLANE, X3 = 64 * T3::PLANE, X4 = 64 * T4::PLANE;
    // conv 3 on 4 item groups x 3 filter rows: pixel blocks rounded up to a multiple of 2 so that the 8 | 12 items divide
    static constexpr int NPB3 = ((H3 * H3 + 31) / 32 + 1) / 2 * 2, IPW3 = NPB3 * 2 / 4;
    static constexpr int RED3 = 2 * 4 * IPW3 * 1024, RED4 = 2 * 4 * 1 * 1024, RED5 = 5 * 2 * 1 * 1024;
    static constexpr int A = ((X2 > RED3 ? X2 : RED3) + 3) & ~3;
    static constexpr int Bsz = ((X3 > RED5 ? X3 : RED5) + 3) & ~3;
    static constexpr int BIA = A + Bsz;                      // biases of the three layers (3 x 64)
    static constexpr int A5S = BIA + 192;                    // act5 of the image, flatten order (576)
    static constexpr int HPS = A5S + kImgFeat;               // head: 6 K-slice partials of the 128 hidden units, then hidden
    static constexpr int LDS_FLOATS = HPS + 6 * kHid + kHid;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(X4 <= A && RED4 <= Bsz, "aliasing plan");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU: the whole LDS");
};

PH_DECL();
}  // namespace
#ifdef VAR_PHASES
extern "C" int var_debug_phases_mid(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
namespace {

template <class C, int H2>
__global__ void __launch_bounds__(MID_NT)
img_fwd_mid_kernel(const float* __restrict__ x2, const float* __restrict__ w3, const float* __restrict__ b3,
                   const float* __restrict__ w4, const float* __restrict__ b4, const float* __restrict__ w5,
                   const float* __restrict__ b5, float* __restrict__ y3, float* __restrict__ y4, float* __restrict__ y5,
                   const float* __restrict__ hw0t, const float* __restrict__ hb0, const float* __restrict__ hw1,
                   float* __restrict__ hid, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    float* ra = lds;
    float* rb = lds + C::A;
    PH_INIT(9);
    if (tid < 64) { lds[C::BIA + tid] = b3[tid]; lds[C::BIA + 64 + tid] = b4[tid]; lds[C::BIA + 128 + tid] = b5[tid]; }
    // region B = conv 4's input tile: zero it once (pads stay zero, data cells are written by conv 3's epilogue)
    lds_zero<MID_NT>(rb, C::Bsz, tid);
    // conv 3 input: the image's act2 planes, padding materialised
    lds_zero_cols<MID_NT>(ra, 32 * C::T2::IR, C::T2::PW, 0, 1, tid);
    lds_zero_cols<MID_NT>(ra, 32 * C::T2::IR, C::T2::PW, H2 + 1, C::T2::PW - H2 - 1, tid);
    PH(0);
    stage_x_band<32, H2, H2, C::T2::IR, C::T2::PW, C::T2::PLANE, false, MID_NT>(ra, x2 + (size_t)b * 32 * H2 * H2, -1, true, tid);
    PH(1);
    __syncthreads();
    PH(2);
    mid_layer<32, 64, H2, 4, 3, 1, C::T3::PLANE, C::T3::PW, C::NPB3>(ra, ra, rb, w3, lds + C::BIA, y3 + (size_t)b * 64 * C::H3 * C::H3, tid);
    PH(3);
    // region A is dead (conv 3's fold has been read): it becomes conv 5's input tile
    __syncthreads();
    lds_zero<MID_NT>(ra, (C::X4 + 3) & ~3, tid);
    __syncthreads();
    PH(4);
    mid_layer16<64, 64, C::H3, 12, 1, C::T4::PLANE, C::T4::PW>(rb, rb, ra, w4, lds + C::BIA + 64, y4 + (size_t)b * 64 * C::H4 * C::H4, tid);
    PH(5);
    __syncthreads();
    PH(6);
    mid_layer16<64, 64, C::H4, 4, 3, 1, 1>(ra, rb, nullptr, w5, lds + C::BIA + 128, y5 + (size_t)b * 64 * C::H5 * C::H5, tid,
                                          lds + C::A5S);
    // ---- image head of this image (imgTriplet, arm_pretext_model.py:46-50): hidden = relu(W0 a5 + b0) on the VALU
    //      (one row: nothing for the matrix cores), 768 threads = 128 hidden units x 6 K slices of 96, folded in
    //      fixed order; then this image's partial of the 128 -> 3 layer in the (row, 4, 4) layout the finish / rows
    //      kernels read (block 0 carries the whole dot product, blocks 1..3 are zero).
    PH(7);
    if (hw0t) {
        static_assert(C::H5 * C::H5 * 64 == kImgFeat && MID_NT == 6 * kHid, "head phase shape");
        __syncthreads();
        const float* a5 = lds + C::A5S;
        float* hp = lds + C::HPS;
        const int j = tid & (kHid - 1), sl = tid >> 7;
        const float* wj = hw0t + (size_t)(sl * 96) * kHid + j;
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < 96; ++k) acc += wj[(size_t)k * kHid] * a5[sl * 96 + k];
        hp[sl * kHid + j] = acc;
        __syncthreads();
        float* hh = hp + 6 * kHid;
        if (tid < kHid) {
            float v = ((hp[tid] + hp[kHid + tid]) + (hp[2 * kHid + tid] + hp[3 * kHid + tid])) +
                      (hp[4 * kHid + tid] + hp[5 * kHid + tid]) + hb0[tid];
            v = v > 0.f ? v : 0.f;
            hh[tid] = v;
            hid[(size_t)b * kHid + tid] = v;
        }
        __syncthreads();
        if (tid < 192) {                                       // 3 outputs x 64 lanes, 2 hidden units per lane
            const int d = tid >> 6, l = tid & 63;
            float sum = hh[l] * hw1[d * kHid + l] + hh[64 + l] * hw1[d * kHid + 64 + l];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
            if (l == 0) part[(size_t)b * 16 + d] = sum;
        }
        if (tid >= 192 && tid < 192 + 13) {                    // the rest of the row: zeros
            const int e = tid - 192 + 3;
            part[(size_t)b * 16 + e] = 0.f;
        }
    }
    PH(8);
}
}  // namespace

// conv 3 + conv 4 + conv 5 of the image CNN (act2 21 x 21 for 84 x 84 inputs, 24 x 24 for 96 x 96); leaves act[3],
// act[4], act[5] and, with_head, the image head's hidden layer (hid_i) and 128 -> 3 partials (head_part rows [0, B))
template <int H2>
static int launch_mid(var_ctx* c, hipStream_t s, const float* params, int B, bool with_head) {
    using C = MidCfg<H2>;
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_fwd_mid_kernel<C, H2>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    hipLaunchKernelGGL((img_fwd_mid_kernel<C, H2>), dim3(B), dim3(MID_NT), C::LDS_BYTES, s, c->act[2],
                       c->wpack + K.img_f[2], params + L.img_b[2], c->wpack + K.img_f[3], params + L.img_b[3],
                       c->wpack + K.img_f[4], params + L.img_b[4], c->act[3], c->act[4], c->act[5],
                       with_head ? c->wpack + K.ih_w0t : nullptr, params + L.ih_b0, params + L.ih_w1, c->hid_i, c->head_part);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int launch_img_fwd_mid(var_ctx* c, hipStream_t s, const float* params, int B, bool with_head) {
    ProfScope prof(c, s, TAG_IMG_FWD0 + 2);
    return c->H == 84 ? launch_mid<21>(c, s, params, B, with_head) : launch_mid<24>(c, s, params, B, with_head);
}
