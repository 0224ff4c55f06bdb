// Weight gradients of image conv 3, 4 and 5 at 84 x 84 in ONE grid (reference: the backward of
// models/pretext/arm_pretext_model.py:12-17 under loss.backward(), VAR/pretext_VAR.py:66):
//   dW[n][c][ky][kx] = sum over images and output pixels of  gy[n][oy][ox] * x[c][2 oy + ky - 1][2 ox + kx - 1],   db[n] = sum gy[n].
//
// One 16-wave workgroup per CU keeps the WHOLE gradient of its layer in registers (conv 3: 64 x 32 x 9, conv 4 / 5:
// 64 x 64 x 9 = 36 accumulator registers per lane) and walks its share of the batch; the three layers split the 256 CUs by
// their work (128 / 96 / 32 workgroups at batch 256: 2 / 3 / 8 images each).  Per image (conv 3: per half image) both operands
// are laid into LDS by LDS-DMA, one dword per lane (global_load_lds_dword gathers: the request needs no registers and stays
// in flight across the barrier), into a two-slot ring: unit i+1 lands while unit i's products run -- one barrier per unit.
//   * gy tile  [n][p], p = flat output pixel of the band, plane pitch == 2 (mod 32)
//   * x tile   [c][row][odd columns -1, 1, ... | even columns 0, 2, ...], plane pitch == 2 (mod 32): with the columns split
//     by parity the four pixels of a k-step (consecutive output pixels) read consecutive words for every tap, so the
//     16 (channel) x 4 (pixel) operand read is conflict-free, and a tap is an immediate offset of the ds_read
//     (ky * row pitch + {0, #odd, 1}).  The zero border (row / column -1 and H) is never written by the DMA.
// v_mfma_f32_16x16x4_f32: D[n][c] += A[n][4 pixels] B[4 pixels][c], one (16 n, 16 c) pair per wave and nine accumulators
// (taps); conv 3 has eight such pairs: its 16 waves split the k-steps in two halves that are added through LDS at the end.
// Every workgroup writes one partial slab per 32 x 32 channel block in the layout img_wgrad_reduce_kernel folds
// ([n 32][tap 9][c 32] + 32 bias sums).
#include "var_common.h"

namespace {
PH_DECL();
}
#ifdef VAR_PHASES
extern "C" int var_debug_phases_wg345(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
#ifndef VAR_PH_BLOCK
#define VAR_PH_BLOCK 3
#endif
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int up2(int v) { const int p = (v & ~31) + 2; return p >= v ? p : p + 32; }     // smallest p >= v with p == 2 (mod 32)

template <int CIN_, int COUT_, int H_, int NBAND_, int KH_>
struct WCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, HO = (H_ - 1) / 2 + 1, NBAND = NBAND_, KH = KH_;
    static constexpr int RB = (HO + NBAND - 1) / NBAND;        // output rows per band
    static constexpr int XR = 2 * RB + 1;                      // input rows per band (2 oy0 - 1 ..)
    static constexpr int NODD = HO + 1, RP = 2 * HO + 1;       // odd columns -1 .. 2 HO - 1, then even columns 0 .. 2 HO - 2
    static constexpr int PX = up2(XR * RP), PY = up2(RB * HO);
    static constexpr int NSUB = COUT / 2;                      // a workgroup owns half of the output channels (one slab per 32 x 32 block)
    static constexpr int XT = CIN * PX, YT = NSUB * PY;
    static constexpr int NIX = (XT + 63) / 64, NIY = (YT + 63) / 64;      // DMA instructions (64 dwords each) per unit
    static constexpr int STAGE = (NIX + NIY) * 64;
    static constexpr int CT = CIN / 16, PAIRS = (NSUB / 16) * CT;
    static constexpr int CBLK = CIN / 32, NCOMBO = (COUT / 32) * CBLK;
    static constexpr int RED = (KH - 1) * PAIRS * 37 * 64;     // accumulators of the other k-slices, folded through LDS at the end
    static_assert(PAIRS * KH == 16 && NSUB == 32, "16 waves; a workgroup owns one 32-channel block of outputs");
    static_assert(2 * HO - 1 <= H_ && RB * NBAND >= HO, "geometry");
};
using L2 = WCfg<32, 64, 21, 2, 4>;       // conv 3: act2 (32, 21, 21) -> (64, 11, 11), two bands of 6 / 5 output rows; 4 pairs x 4 k-slices
using L3 = WCfg<64, 64, 11, 1, 2>;       // conv 4: act3 (64, 11, 11) -> (64, 6, 6); 8 pairs x 2 k-slices
using L4 = WCfg<64, 64, 6, 1, 2>;        // conv 5: act4 (64, 6, 6) -> (64, 3, 3)
constexpr int kSlab = 9248;
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int kStageMax = cmax(L2::STAGE, cmax(L3::STAGE, L4::STAGE));
constexpr int kRed = cmax(L2::RED, cmax(L3::RED, L4::RED));
constexpr int kLdsFloats = cmax(2 * kStageMax, kRed) + 64;      // + the always-zero cell (A operand of a padding pixel)
constexpr int kZero = cmax(2 * kStageMax, kRed);

template <class C>
__device__ __forceinline__ void wg_body(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ slabs, int B,
                                        int g, int G, int nh, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave % C::PAIRS, kh = wave / C::PAIRS;
    const int nt = pair / C::CT, ct = pair % C::CT;
    const int ncol = lane & 15, kq = lane >> 4;
    const int nimg = g < B ? (B - g + G - 1) / G : 0;
    const int nunits = nimg * C::NBAND;

    // ---- staging: global -> registers (in flight during the previous unit's products) -> LDS.  Cell e = tid + 1024 i of the
    // unit's tile takes element goff of the image; the (cell -> element) map is the same for every unit up to the band's row
    // shift, so it is decoded once into one table word per cell: goff | row << 16 | (column inside the image) << 21.
    constexpr int NXI = (C::XT + 1023) / 1024, NYI = (C::YT + 1023) / 1024;
    unsigned tx[NXI], ty[NYI];
    float rx[NXI], ry[NYI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int e = tid + 1024 * i;
        const int c = e / C::PX, r = e - c * C::PX;
        const int row = r / C::RP, q = r - row * C::RP;
        const int col = q < C::NODD ? 2 * q - 1 : 2 * (q - C::NODD);
        const bool ok = e < C::XT && row < C::XR && col >= 0 && col < C::H;
        tx[i] = ok ? (unsigned)((c * C::H + row) * C::H + col) | ((unsigned)row << 16) | (1u << 21) : 0u;
    }
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
        const int e = tid + 1024 * i;
        const int n = e / C::PY, p = e - n * C::PY;
        ty[i] = e < C::YT ? (unsigned)((nh * C::NSUB + n) * C::HO * C::HO + p) | ((unsigned)p << 16) | (1u << 24) : 0u;
    }
    auto gload = [&](int u) {
        const int b = g + (u / C::NBAND) * G, band = u % C::NBAND;
        const int oy0 = band * C::RB;
        const int rows = C::HO - oy0 < C::RB ? C::HO - oy0 : C::RB;
        const int npx = rows * C::HO, iy0 = 2 * oy0 - 1;
        const float* xi = x + (size_t)b * C::CIN * C::H * C::H + iy0 * C::H;
        const float* yi = gy + (size_t)b * C::COUT * C::HO * C::HO + oy0 * C::HO;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int iy = iy0 + (int)((tx[i] >> 16) & 31);
            rx[i] = ((tx[i] >> 21) && iy >= 0 && iy < C::H) ? xi[tx[i] & 0xffff] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NYI; ++i) ry[i] = ((ty[i] >> 24) && (int)((ty[i] >> 16) & 255) < npx) ? yi[ty[i] & 0xffff] : 0.f;
    };
    auto lstore = [&](int u) {
        float* st = lds + (u & 1) * C::STAGE;
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            if (tid + 1024 * i < C::XT) st[tid + 1024 * i] = rx[i];
#pragma unroll
        for (int i = 0; i < NYI; ++i)
            if (tid + 1024 * i < C::YT) st[C::NIX * 64 + tid + 1024 * i] = ry[i];
    };

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int a_lane = C::NIX * 64 + (nt * 16 + ncol) * C::PY;      // + stage + p
    const int b_lane = (ct * 16 + ncol) * C::PX;                    // + stage + 2 oy RP + ox

    PHR_INIT(VAR_PH_BLOCK, VAR_PH_THREAD);
    if (nunits > 0) { gload(0); lstore(0); }
    __syncthreads();
    PHR(0);
#pragma unroll 1
    for (int u = 0; u < nunits; ++u) {
        // the next unit's loads: half of the waves (two per SIMD) issue them before their products, the other half in the middle
        // of theirs, so that the matrix pipe always has waves to run while the others are busy with addresses
        const bool more = u + 1 < nunits, late = wave >= 8;
        if (more && !late) gload(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        PHR(1);
        const int band = u % C::NBAND, oy0 = band * C::RB;
        const int rows = C::HO - oy0 < C::RB ? C::HO - oy0 : C::RB;
        const int npx = rows * C::HO, nstep = (npx + 3) >> 2;
        const int sbase = (u & 1) * C::STAGE;
        auto addr = [&](int s, int& ao, int& bo) {
            const int p = 4 * s + kq;
            const bool v = p < npx;
            const int pc = v ? p : npx - 1;
            const int oy = pc / C::HO, ox = pc - oy * C::HO;
            ao = v ? sbase + a_lane + p : kZero;
            bo = sbase + b_lane + 2 * oy * C::RP + ox;
        };
        auto load = [&](int ao, int bo, float& a, float (&b)[9]) {
            a = lds[ao];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t % 3;
                b[t] = lds[bo + ky * C::RP + (kx == 0 ? 0 : kx == 1 ? C::NODD : 1)];
            }
        };
        auto mma = [&](float a, const float (&b)[9]) {
            bsum += a;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
        };
        // two operand sets: the reads of the next step are issued before this step's nine products (unconditionally -- past the
        // end they re-read the last step -- so that the loop body is branch-free straight-line code for the scheduler)
        auto steps = [&](int sb, int se) {                          // this wave's steps (kh, kh + KH, ...) of [sb, se), sb a multiple of KH
            float a0, a1, b0[9], b1[9];
            int ao, bo;
            int s = sb + kh;
            if (s >= se) return;
            addr(s, ao, bo); load(ao, bo, a0, b0);
#pragma unroll 1
            for (;;) {
                const int s2 = s + C::KH;
                addr(s2 < se ? s2 : se - 1, ao, bo); load(ao, bo, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (s2 >= se) break;
                s = s2 + C::KH;
                addr(s < se ? s : se - 1, ao, bo); load(ao, bo, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                if (s >= se) break;
            }
        };
        const int smid = (nstep / (2 * C::KH)) * C::KH;
        steps(0, smid);
        if (more && late) gload(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        steps(smid, nstep);
        PHR(2);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < nunits) lstore(u + 1);                          // the other slot: nobody reads it during unit u
        PHR(3);
        __syncthreads();
        PHR(4);
    }

    // ---- the other k-slices join the first through LDS (fixed order), then one partial slab per (32 n, 32 c) block ----
    if constexpr (C::KH > 1) {
        __syncthreads();
        if (kh > 0) {
            float* red = lds + ((kh - 1) * C::PAIRS + pair) * 37 * 64 + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64] = acc[t][r];
            red[36 * 64] = bsum;
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll 1
            for (int q = 0; q < C::KH - 1; ++q) {
                const float* red = lds + (q * C::PAIRS + pair) * 37 * 64 + lane;
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] += red[(t * 4 + r) * 64];
                bsum += red[36 * 64];
            }
        }
    }
    if (kh == 0 && g < B) {
        const int cb = ct >> 1;                                     // (NSUB == 32: this workgroup's n-block is nh)
        float* slab = slabs + ((size_t)g * C::NCOMBO + nh * C::CBLK + cb) * kSlab;
        const int c32 = (ct & 1) * 16 + ncol;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n32 = (nt & 1) * 16 + 4 * kq + r;
                slab[(n32 * 9 + t) * 32 + c32] = acc[t][r];
            }
        if (ct == 0) {
            float bs = bsum + __shfl_xor(bsum, 16, 64);
            bs += __shfl_xor(bs, 32, 64);
            if (kq == 0) slab[9216 + (nt & 1) * 16 + ncol] = bs;
        }
    }
    PHR(5);
    PHR_FLUSH();
}

__global__ void __launch_bounds__(1024)
img_wg345_kernel(const float* __restrict__ x2, const float* __restrict__ x3, const float* __restrict__ x4,
                 const float* __restrict__ gy3, const float* __restrict__ gy4, const float* __restrict__ gy5,
                 float* __restrict__ slabs2, float* __restrict__ slabs3, float* __restrict__ slabs4, int G2, int G3, int G4, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < kLdsFloats; i += 4096) *(f32x4*)(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    int id = blockIdx.x;                                            // (layer, image group g of G, output-channel half): longest first
    if (id < 2 * G2) { wg_body<L2>(x2, gy3, slabs2, B, id >> 1, G2, id & 1, lds); return; }
    id -= 2 * G2;
    if (id < 2 * G3) { wg_body<L3>(x3, gy4, slabs3, B, id >> 1, G3, id & 1, lds); return; }
    id -= 2 * G3;
    wg_body<L4>(x4, gy5, slabs4, B, id >> 1, G4, id & 1, lds);
}
}  // namespace

int launch_img_wg345(var_ctx* c, hipStream_t s, int B) {
    ProfScope prof(c, s, TAG_IMG_WGRAD0 + 2);
    static_assert(kLdsFloats % 4 == 0, "vector zeroing");
    constexpr int LDS_BYTES = kLdsFloats * 4;
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_wg345_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    auto groups = [&](int layer) { const int cap = img_wgrad_groups84(layer); return B < cap ? B : cap; };
    const int G2 = groups(2), G3 = groups(3), G4 = groups(4);
    c->wg_groups[2] = G2; c->wg_groups[3] = G3; c->wg_groups[4] = G4;
    hipLaunchKernelGGL(img_wg345_kernel, dim3(2 * (G2 + G3 + G4)), dim3(1024), LDS_BYTES, s, c->act[2], c->act[3], c->act[4], c->gact[3],
                       c->gact[4], c->gact[5], c->slabs + img_slab_offset(2), c->slabs + img_slab_offset(3),
                       c->slabs + img_slab_offset(4), G2, G3, G4, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
