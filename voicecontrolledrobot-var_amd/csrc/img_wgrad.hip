// Image CNN weight gradients (autograd of models/pretext/arm_pretext_model.py:9-18):
//   dW[n][c][tap] = sum_{b,oy,ox} gy[b][n][oy][ox] * x[b][c][2oy+ky-1][2ox+kx-1],  db[n] = sum gy[b][n][..]
// on the gfx950 f32 matrix cores:  D[n][c] per tap, reduction index = output pixels.
//
// Work decomposition
//   blockIdx.y = (32-channel block of n, 32-channel block of c); a workgroup has 12 waves = 3 ky x 4 K slices,
//   each wave owns the three kx tiles of its ky (A operand shared by the three MFMAs) over its slice of
//   the staged pixels; blockIdx.x strides over "units" (R output rows of one image): split-K across
//   workgroups.  The first layer (CIN = 3) packs (tap, c) into one 32-wide tile: 8 waves = 8 K slices.
// Software pipeline (the point of this kernel)
//   (ColStager, img_stage.h) every lane issues ALL global loads of the NEXT unit into registers right after the barrier
//   that precedes the MFMAs of the CURRENT unit, and writes them to LDS after those MFMAs:
//   HBM/L2 latency is overlapped with matrix-core work instead of being paid per batch of loads.
// Determinism
//   each workgroup writes one partial slab; img_wgrad_reduce_kernel sums slabs in a fixed order.
#include <stdlib.h>

#include <type_traits>

#include "img_stage.h"

template <int CIN_, int COUT_, int H_, bool U8_, int R_, int NU_, int KS_, bool DB_ = false>
struct WgCfg {
    static constexpr bool DB = DB_;                 // two LDS stages: the next unit is stored while this one is multiplied
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, R = R_, NU = NU_;
    static constexpr bool U8 = U8_;
    static constexpr bool SMALLC = (CIN < 32);      // first layer: columns = (tap, c), 27 of 32 used
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int IR = 2 * R + 1;
    static constexpr int PW = 2 * WO + 3;
    static constexpr int XC = SMALLC ? CIN : 32;    // x channels staged by one workgroup
    static constexpr int PLANE_X = (IR * PW) | 1;   // odd: lanes differ in channel
    static constexpr int UNIT_X = XC * PLANE_X;
    static constexpr int POW = WO + 1;              // zero column at ox = WO
    static constexpr int PLANE_Y = (R * POW) | 1;
    static constexpr int UNIT_Y = 32 * PLANE_Y;
    static constexpr int NB = (HO + R - 1) / R;
    static constexpr int NBLK = COUT / 32;
    static constexpr int CBLK = SMALLC ? 1 : CIN / 32;
    static constexpr int NCOMBO = NBLK * CBLK;
    static constexpr int KS = KS_;                  // K (pixel) slices inside the workgroup
    static constexpr int NW = SMALLC ? KS : 3 * KS; // wave = ks*3 + ky
    static constexpr int NT = NW * 64;
    static constexpr int XS = (NU * UNIT_X + 3) & ~3, YS = NU * UNIT_Y;
    static constexpr int ZPAD = (XS + YS + 3) & ~3;  // a few always-zero floats: A operand of idle k-steps
    static constexpr int STAGE = ZPAD + 4;
    static constexpr int FOLD = SMALLC ? NW * 1024 : NW * 3 * 1024;     // every wave parks its tiles at once (one round, one barrier pair)
    static constexpr int LDS_FLOATS = ((DB ? 2 : 1) * STAGE > FOLD ? (DB ? 2 : 1) * STAGE : FOLD);
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int SLAB = SMALLC ? (32 * 32 + 32) : (32 * 9 * 32 + 32);
    static constexpr int HSTEPS = (WO + 1) / 2;
    static constexpr int TS = NU * R * HSTEPS;      // k-steps per stage
    static constexpr int NSTEP = (TS + KS - 1) / KS; // k-steps per wave and stage
};

// (a __device__ body so that it can also run as one half of a fused launch, img_conv_bwd.hip: bx / by / G stand for
// blockIdx.x / blockIdx.y / gridDim.x of a stand-alone launch)
#ifndef VAR_WG_PH
#define VAR_WG_PH
namespace { PH_DECL(); }
#if defined(VAR_PHASES) && defined(VAR_WGRAD_DEVICE_ONLY)      // (the entry lives in img_conv_bwd.o, whose grids run these bodies at 84 x 84)
extern "C" int var_debug_phases_wgrad(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
#endif
#ifndef VAR_WG_PH_BLOCK
#define VAR_WG_PH_BLOCK 0
#endif
template <class C>
__device__ __forceinline__ void img_wgrad_body(const void* __restrict__ xin, long bstride, const int* __restrict__ bidx,
                                               const float* __restrict__ gy, float* __restrict__ slabs, int B,
                                               int bx, int by, int G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ys = lds + C::XS;
    constexpr int NT = C::NT;
    using XT = typename std::conditional<C::U8, uint8_t, float>::type;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int combo = by;
    const int nb = combo / C::CBLK, cb = combo - nb * C::CBLK;
    const int ky = C::SMALLC ? 0 : wave % 3, ks = C::SMALLC ? wave : wave / 3;

    // whole odd-width maps (84 x 84: the 21 x 21 and 11 x 11 layers) are read as float4 runs (FlatStager, img_stage.h)
    constexpr bool FLATX = !C::U8 && !C::SMALLC && C::NB == 1 && (C::W & 1) && C::R == C::HO;
    constexpr bool FLATY = !C::SMALLC && C::NB == 1 && (C::WO & 1) && C::R == C::HO;
    typename std::conditional<FLATX, FlatStager<C::XC, C::H, C::W, C::PW, C::PLANE_X, 1, 1, NT, C::NU>,
                              ColStager<C::XC, C::H, C::W, C::IR, C::PW, C::PLANE_X, 1, C::U8, NT, C::NU>>::type sx;
    typename std::conditional<FLATY, FlatStager<32, C::HO, C::WO, C::POW, C::PLANE_Y, 0, 0, NT, C::NU>,
                              ColStager<32, C::HO, C::WO, C::R, C::POW, C::PLANE_Y, 0, false, NT, C::NU>>::type sy;
    sx.init(tid);
    sy.init(tid);

    static_assert(C::NU == 1 || C::NB == 1, "multi-unit stages need whole-image units (consecutive images)");
    // stage bookkeeping (wave-uniform): first unit -> image pointers, band rows, number of live units
    struct Stage { const XT* bx; const float* by; int row0x, row0y, nvalid; };
    auto make_stage = [&](int unit0) {
        Stage st;
        const int bo = unit0 / C::NB, band = unit0 - bo * C::NB;
        const int b = bidx ? bidx[bo] : bo;                      // optional batch gather (first layer, NU == 1)
        st.bx = (const XT*)xin + (size_t)b * bstride + (size_t)(cb * C::XC) * C::H * C::W;
        st.by = gy + ((size_t)bo * C::COUT + nb * 32) * C::HO * C::WO;
        st.row0x = 2 * band * C::R - 1;
        st.row0y = band * C::R;
        st.nvalid = total_units - unit0 < C::NU ? total_units - unit0 : C::NU;
        return st;
    };
    constexpr long YSTRIDE = (long)C::COUT * C::HO * C::WO;

    // ---- lane offsets for the MFMA operands ----
    const int aoff = l31 * C::PLANE_Y + half;                       // + u*UNIT_Y + oyl*POW + 2s
    int boff;
    if constexpr (C::SMALLC) {
        const int col = l31 < C::CIN * 9 ? l31 : 0;                  // col = tap*CIN + c
        const int tap = col / C::CIN, c = col - tap * C::CIN;
        boff = c * C::PLANE_X + (tap / 3) * C::PW + (tap % 3) + 2 * half;
    } else {
        boff = l31 * C::PLANE_X + ky * C::PW + 2 * half;             // + u*UNIT_X + 2*oyl*PW + 4s + kx
    }
    // This wave's K slice as a fixed list of LDS offsets (computed once): the stage loop below is then
    // straight-line code and the compiler can hoist the operand reads of later steps above the MFMAs
    // of earlier ones.  Idle steps (uneven split) read A from an always-zero cell.
    const int t0 = (ks * C::TS) / C::KS, t1 = ((ks + 1) * C::TS) / C::KS;
    int aos[C::NSTEP], bos[C::NSTEP];
#pragma unroll
    for (int i = 0; i < C::NSTEP; ++i) {
        const int t = t0 + i;
        const bool live = t < t1;
        const int tt = live ? t : t0;
        const int u = tt / (C::R * C::HSTEPS), rem = tt - u * (C::R * C::HSTEPS);
        const int oyl = rem / C::HSTEPS, sstep = rem - oyl * C::HSTEPS;
        aos[i] = live ? C::XS + aoff + u * C::UNIT_Y + oyl * C::POW + 2 * sstep : C::ZPAD;
        bos[i] = boff + u * C::UNIT_X + 2 * oyl * C::PW + 4 * sstep;
    }
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    PHR_INIT(VAR_WG_PH_BLOCK, VAR_PH_THREAD);
    // zero the pads once (data cells are rewritten for every unit)
    lds_zero<NT>(lds, C::LDS_FLOATS, tid);
    const int first = bx * C::NU;
    if (first < total_units) {
        const Stage st = make_stage(first);
        sx.issue(st.bx, bstride, st.row0x, st.nvalid);
        sy.issue(st.by, YSTRIDE, st.row0y, st.nvalid);
    }
    auto multiply = [&](const float* stage) {
#pragma unroll
        for (int i = 0; i < C::NSTEP; ++i) {
            const float a = stage[aos[i]];
            bsum += a;
            if constexpr (C::SMALLC) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, stage[bos[i]], acc[0], 0, 0, 0);
            } else {
                const float b0 = stage[bos[i]], b1 = stage[bos[i] + 1], b2 = stage[bos[i] + 2];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
            }
        }
    };
    if constexpr (C::DB) {
        // Two stages: unit i+1 is written into the other stage right after this wave's MFMAs of unit i, while
        // slower waves are still multiplying -- one barrier per unit, and the LDS store pass hides behind matrix work.
        __syncthreads();
        if (first < total_units) {
            const Stage st = make_stage(first);
            sx.store(lds, st.row0x, st.nvalid);
            sy.store(lds + C::XS, st.row0y, st.nvalid);
        }
        __syncthreads();
        int it = 0;
#pragma unroll 1
        for (int unit0 = first; unit0 < total_units; unit0 += G * C::NU, ++it) {
            const int next = unit0 + G * C::NU;
            const bool more = next < total_units;
            const Stage sn = make_stage(more ? next : unit0);
            if (more) {
                sx.issue(sn.bx, bstride, sn.row0x, sn.nvalid);
                sy.issue(sn.by, YSTRIDE, sn.row0y, sn.nvalid);
            }
            __builtin_amdgcn_sched_barrier(0);
            multiply(lds + (it & 1) * C::STAGE);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                float* nst = lds + ((it + 1) & 1) * C::STAGE;
                sx.store(nst, sn.row0x, sn.nvalid);
                sy.store(nst + C::XS, sn.row0y, sn.nvalid);
            }
            __syncthreads();
        }
    } else {
        PHR(0);
#pragma unroll 1
        for (int unit0 = first; unit0 < total_units; unit0 += G * C::NU) {
            __syncthreads();                              // previous stage's MFMAs are done reading LDS
            PHR(1);
            {
                const Stage st = make_stage(unit0);       // same (cheap, uniform) bookkeeping the loads were issued with
                sx.store(xs, st.row0x, st.nvalid);
                sy.store(ys, st.row0y, st.nvalid);
            }
            PHR(2);
            __syncthreads();
            PHR(3);
            if (unit0 + G * C::NU < total_units) {        // in flight during the MFMAs below
                const Stage st = make_stage(unit0 + G * C::NU);
                sx.issue(st.bx, bstride, st.row0x, st.nvalid);
                sy.issue(st.by, YSTRIDE, st.row0y, st.nvalid);
            }
            PHR(4);
            __builtin_amdgcn_sched_barrier(0);
            multiply(lds);
            __builtin_amdgcn_sched_barrier(0);
            PHR(5);
        }
    }

    // ---- fold the K slices through LDS (fixed order) and write this workgroup's partial slab ----
    float* slab = slabs + ((size_t)bx * C::NCOMBO + combo) * C::SLAB;
    bsum += __shfl_down(bsum, 32, 64);
    if constexpr (C::SMALLC) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
            lds[wave * 1024 + n * 32 + l31] = acc[0][r];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += NT) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < C::KS; ++q) sum += lds[q * 1024 + e];
            slab[e] = sum;
        }
    } else {
        // all three kx tiles of every wave in ONE round (was three, two barriers each); four consecutive c per thread and one
        // 16-byte store
        __syncthreads();
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                lds[(wave * 3 + kx) * 1024 + n * 32 + l31] = acc[kx][r];
            }
        __syncthreads();
        for (int e = tid; e < 9 * 256; e += NT) {
            const int tap = e >> 8, kyy = tap / 3, kx = tap - 3 * kyy, i = (e & 255) * 4;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < C::KS; ++q) sum += *(const f32x4*)(lds + ((q * 3 + kyy) * 3 + kx) * 1024 + i);
            *(f32x4*)(slab + ((i >> 5) * 9 + tap) * 32 + (i & 31)) = sum;
        }
    }
    __syncthreads();
    if (ky == 0 && half == 0) lds[ks * 32 + l31] = bsum;
    __syncthreads();
    if (tid < 32) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < C::KS; ++q) sum += lds[q * 32 + tid];
        slab[(C::SMALLC ? 1024 : 9216) + tid] = sum;
    }
    PHR(6);
    PHR_FLUSH();
}

//                   CIN COUT  H   U8    R  NU KS
using W84_2 = WgCfg<32, 64, 21, false, 11, 1, 4>;
using W84_3 = WgCfg<64, 64, 11, false, 6, 4, 4>;       // four images per stage: a unit of this layer is too short to cover a load round trip
using W84_4 = WgCfg<64, 64, 6, false, 3, 4, 4>;
using W96_2 = WgCfg<32, 64, 24, false, 6, 1, 4>;
using W96_3 = WgCfg<64, 64, 12, false, 6, 1, 4>;
using W96_4 = WgCfg<64, 64, 6, false, 3, 4, 4>;

#ifndef VAR_WGRAD_DEVICE_ONLY      // img_conv_bwd.hip includes this file for the device code above only
// ------------------------------------------------------------------------------------------
// slab reduction -> OIHW gradient arena
// ------------------------------------------------------------------------------------------
struct RedSeg { int slab_off; int slab_sz; int G; int ncombo; int cblk; int cin; int smallc; int gw; int gb; };
struct RedTable { RedSeg seg[5]; int start[6]; };

// 32 consecutive slab elements x 8 slices of the workgroup index per block: each lane sums every 8th
// slab (loads unrolled), the 8 partial sums are folded in a fixed order through LDS.
__global__ void __launch_bounds__(256)
img_wgrad_reduce_kernel(RedTable T, const float* __restrict__ slabs, float* __restrict__ grads) {
    __shared__ float part[8][33];
    const int lane32 = threadIdx.x & 31, gs = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + lane32;
    const bool live = j < T.start[5];
    int l = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) if (j >= T.start[i] && T.start[i + 1] > T.start[i]) l = i;
    const RedSeg S = T.seg[l];
    const int e = live ? j - T.start[l] : 0;         // element of the (combo, slab) space of this layer
    const float* p = slabs + S.slab_off + e;
    const size_t gstride = (size_t)S.ncombo * S.slab_sz;
    float s = 0.f;
    if (live) {
#pragma unroll 8
        for (int g = gs; g < S.G; g += 8) s += p[(size_t)g * gstride];
    }
    part[gs][lane32] = s;
    __syncthreads();
    if (gs != 0 || !live) return;
    s = ((part[0][lane32] + part[1][lane32]) + (part[2][lane32] + part[3][lane32])) +
        ((part[4][lane32] + part[5][lane32]) + (part[6][lane32] + part[7][lane32]));
    const int combo = e / S.slab_sz, i = e - combo * S.slab_sz;
    if (S.smallc) {
        if (i >= 1024) { grads[S.gb + (i - 1024)] = s; return; }
        const int n = i / 32, col = i % 32;
        if (col < S.cin * 9) {
            const int tap = col / S.cin, c = col - tap * S.cin;
            grads[S.gw + (n * S.cin + c) * 9 + tap] = s;
        }
        return;
    }
    const int nb = combo / S.cblk, cb = combo - nb * S.cblk;
    if (i >= 9216) {
        if (cb == 0) grads[S.gb + nb * 32 + (i - 9216)] = s;
        return;
    }
    const int c = i % 32, tap = (i / 32) % 9, n = i / 288;
    grads[S.gw + ((nb * 32 + n) * S.cin + cb * 32 + c) * 9 + tap] = s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// split-K workgroups (grid.x) per layer; grid.y = channel-block combos.  Also sizes the slab workspace.
static const int kWgG[5] = {512, 256, 64, 32, 32};     // layer 0: the fused tail (kTailG <= 512)
static const int kCombo[5] = {1, 1, 2, 4, 4};
static const int kSlabSz[5] = {32 * 32 + 32, 9248, 9248, 9248, 9248};

size_t img_slab_floats() {
    size_t t = 0;
    for (int i = 0; i < 5; i++) t += (size_t)kWgG[i] * kCombo[i] * kSlabSz[i];
    return t;
}

static size_t slab_offset(int layer) {
    size_t o = 0;
    for (int i = 0; i < layer; i++) o += (size_t)kWgG[i] * kCombo[i] * kSlabSz[i];
    return o;
}

// fixed-order slab sums of layers [lo, hi] into the gradient arena
int launch_img_wgrad_reduce(var_ctx* c, hipStream_t s, float* grads, int lo, int hi) {
    const ParamLayout& L = c->pl;
    RedTable T{};
    int st = 0;
    for (int i = 0; i < 5; i++) {
        T.seg[i] = RedSeg{(int)slab_offset(i), kSlabSz[i], c->wg_groups[i], kCombo[i], i < 3 ? 1 : 2, kImgCh[i],
                          i == 0 ? 1 : 0, L.img_w[i], L.img_b[i]};
        T.start[i] = st;
        if (i >= lo && i <= hi) st += kCombo[i] * kSlabSz[i];
    }
    T.start[5] = st;
    ProfScope prof(c, s, TAG_IMG_WREDUCE);
    hipLaunchKernelGGL(img_wgrad_reduce_kernel, dim3((st + 31) / 32), dim3(256), 0, s, T, c->slabs, grads);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// workgroups per layer / slab workspace offsets, for the fused launches of img_conv_bwd.hip
int img_wgrad_groups(int layer) { return kWgG[layer]; }
size_t img_slab_offset(int layer) { return slab_offset(layer); }
#endif  // VAR_WGRAD_DEVICE_ONLY
