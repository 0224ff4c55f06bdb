// Image CNN backward (autograd of models/pretext/arm_pretext_model.py:9-18 under
// loss.backward(), VAR/pretext_VAR.py:68), hand-written for gfx950 f32 matrix cores.
//
// Three kernels per conv layer  y = relu(conv3x3_s2_p1(x, W) + b):
//   dgrad : gx = (W^T (*) gy) . (x > 0)      -- gradient wrt the previous layer's pre-activation
//           (gy is already masked by y > 0 by whoever produced it).  Stride-2 transposed conv
//           done as 2 x 2 parity classes: an input pixel (iy,ix) receives taps
//           ky in {1} (iy even) or {0,2} (iy odd), same for x, so each class is a dense
//           implicit GEMM  D[c][pixel] = sum_{tap,n} Wd[tap][n][c] * gy[n][pixel shifted].
//           A lane owns the horizontally adjacent pair (ix=2i, ix=2i+1) -> 8-byte stores.
//   wgrad : dW[n][c][tap] = sum_{b,oy,ox} gy[b][n][oy][ox] * x[b][c][2oy+ky-1][2ox+kx-1]
//           D[n][c] per tap, K = pixels; a workgroup walks its units (bands of rows) keeping the
//           accumulators in registers, then writes ONE partial slab; bias sums ride along.
//   reduce: fixed-order sum of the slabs into the OIHW gradient arena (bitwise reproducible;
//           no float atomics).
#include "img_stage.h"

// ------------------------------------------------------------------------------------------
// dgrad
// ------------------------------------------------------------------------------------------
template <int CIN_, int COUT_, int H_, int RI_, int NU_, int NW_>
struct DgCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, RI = RI_, NU = NU_, NW = NW_;
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int NR = RI / 2 + 1;            // gy rows per unit
    static constexpr int POW = WO + 1;               // + zero column at ox = WO
    static constexpr int PLANE = NR * POW;
    static constexpr int UNIT = COUT * PLANE;
    static constexpr int NB = (H + RI - 1) / RI;     // bands per image
    static constexpr int WH = (W + 1) / 2;           // pixel pairs per row
    static constexpr int PPU = (RI / 2) * WH;        // pixel pairs per unit and row-parity class
    static constexpr int NPP = NU * PPU;
    static constexpr int NPB = (NPP + 31) / 32;
    static constexpr int CBLK = CIN / 32;
    static constexpr int ITEMS = NPB * CBLK;         // per class
    static constexpr int IPC = (ITEMS + NW - 1) / NW;
    static constexpr int LDS_FLOATS = (NU * UNIT + 3) / 4 * 4;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(RI % 2 == 0, "band must hold whole row pairs");
};

template <class C>
__global__ void __launch_bounds__(C::NW * 64)
img_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ wd, const float* __restrict__ x,
                 float* __restrict__ gx, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int unit0 = blockIdx.x * C::NU;

    lds_zero<NT>(lds, C::LDS_FLOATS, tid);
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < C::NU; ++u) {
        const int unit = unit0 + u;
        const bool uvalid = unit < total_units;
        const int b = uvalid ? unit / C::NB : 0, band = unit % C::NB;
        stage_y_band<C::COUT, C::HO, C::WO, C::NR, C::POW, C::PLANE, NT>(
            lds + u * C::UNIT, gy + (size_t)b * C::COUT * C::HO * C::WO, band * (C::RI / 2), uvalid, tid);
    }
    __syncthreads();

    const float* wl = wd + half * C::CIN + l31;        // + (tap*COUT + n)*CIN + cb*32
#pragma unroll 1
    for (int ci = 0; ci < C::IPC; ++ci) {
        const int idx = wave + C::NW * ci;
        if (idx >= C::ITEMS) break;
        const int pb = idx % C::NPB, cb = idx / C::NPB;
        int pp = pb * 32 + l31;
        const bool ppvalid = pp < C::NPP;
        if (!ppvalid) pp = 0;
        const int u = pp / C::PPU, q = pp - u * C::PPU;
        const int j = q / C::WH, i = q - j * C::WH;
        const int base = u * C::UNIT + j * C::POW + i + half * C::PLANE;
        const int unit = unit0 + u;
        const int b = unit / C::NB, band = unit - b * C::NB;
        const float* wc = wl + cb * 32;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            // row taps: py=0 -> ky=1 (doy 0);  py=1 -> ky=0 (doy 1), ky=2 (doy 0).
            // Filter values (L2-resident packed image) are prefetched one block of U n-pairs ahead.
            constexpr int U = 8;
            constexpr int BPT = (C::COUT / 2) / U;               // blocks per row tap
            const int NBK = (py ? 2 : 1) * BPT;
            float wb[2][3][U];
            {
                const int ky0 = py ? 0 : 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        wb[0][kx][u] = wc[(size_t)((ky0 * 3 + kx) * C::COUT + 2 * u) * C::CIN];
            }
#pragma unroll
            for (int blk = 0; blk < (py ? 2 : 1) * BPT; ++blk) {
                const int t = blk / BPT, nb0 = (blk % BPT) * U;
                const int doy = (py && !t) ? 1 : 0;
                if (blk + 1 < NBK) {
                    const int t1 = (blk + 1) / BPT, nb1 = ((blk + 1) % BPT) * U;
                    const int ky1 = py ? (t1 ? 2 : 0) : 1;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            wb[(blk + 1) & 1][kx][u] = wc[(size_t)((ky1 * 3 + kx) * C::COUT + 2 * (nb1 + u)) * C::CIN];
                }
                __builtin_amdgcn_sched_barrier(0);      // prefetch loads stay above this block's MFMAs
                const int lb = base + doy * C::POW;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float b0 = lds[lb + 2 * (nb0 + u) * C::PLANE];
                    const float b1 = lds[lb + 2 * (nb0 + u) * C::PLANE + 1];
                    // kx=1 -> px=0 (dox 0); kx=2 -> px=1 (dox 0); kx=0 -> px=1 (dox 1)
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][1][u], b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][2][u], b0, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[blk & 1][0][u], b1, acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: rows = channel c, col (lane&31) = pixel pair
            const int iy = band * C::RI + 2 * j + py;
            if (ppvalid && unit < total_units && iy < C::H) {
                const int ix = 2 * i;
                const size_t o = (size_t)b * C::CIN * C::H * C::W + (size_t)iy * C::W + ix;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const size_t oc = o + (size_t)c * C::H * C::W;
                    if constexpr (C::W % 2 == 0) {
                        const float2 xv = *(const float2*)(x + oc);
                        float2 g;
                        g.x = xv.x > 0.f ? acc0[r] : 0.f;
                        g.y = xv.y > 0.f ? acc1[r] : 0.f;
                        *(float2*)(gx + oc) = g;
                    } else {
                        gx[oc] = x[oc] > 0.f ? acc0[r] : 0.f;
                        if (ix + 1 < C::W) gx[oc + 1] = x[oc + 1] > 0.f ? acc1[r] : 0.f;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------
template <int CIN_, int COUT_, int H_, bool U8_, int R_, int NU_>
struct WgCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, R = R_, NU = NU_;
    static constexpr bool U8 = U8_;
    static constexpr bool SMALLC = (CIN < 32);      // first layer: columns = (tap, c), 27 of 32 used
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int IR = 2 * R + 1;
    static constexpr int PW = 2 * WO + 3;
    static constexpr int PLANE_X = (IR * PW) | 1;   // odd: lanes differ in channel
    static constexpr int UNIT_X = CIN * PLANE_X;
    static constexpr int POW = WO + 1;              // zero column at ox = WO
    static constexpr int PLANE_Y = (R * POW) | 1;
    static constexpr int UNIT_Y = COUT * PLANE_Y;
    static constexpr int NB = (HO + R - 1) / R;
    static constexpr int NBLK = COUT / 32;
    static constexpr int CBLK = SMALLC ? 1 : CIN / 32;
    static constexpr int NW = SMALLC ? 4 : NBLK * CBLK * 3;   // wave -> (nb, cb, ky) | K-split
    static constexpr int XS = (NU * UNIT_X + 3) & ~3, YS = NU * UNIT_Y;
    static constexpr int LDS_FLOATS = ((XS + YS) > (SMALLC ? 4 * 1024 : 0) ? (XS + YS) : 4 * 1024) + 3 & ~3;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int SLAB = SMALLC ? (COUT * 32 + COUT) : (COUT * 9 * CIN + COUT);
    static constexpr int HSTEPS = (WO + 1) / 2;
};

template <class C>
__global__ void __launch_bounds__(C::NW * 64)
img_wgrad_kernel(const void* __restrict__ xin, long bstride, const int* __restrict__ bidx,
                 const float* __restrict__ gy, float* __restrict__ slabs, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ys = lds + C::XS;
    constexpr int NT = C::NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int G = gridDim.x;

    // wave role
    int nb = 0, cb = 0, ky = 0;
    if constexpr (!C::SMALLC) { ky = wave % 3; cb = (wave / 3) % C::CBLK; nb = wave / (3 * C::CBLK); }
    // lane offsets
    int aoff = (nb * 32 + l31) * C::PLANE_Y + half;               // + u*UNIT_Y + oyl*POW + ox
    int boff;
    if constexpr (C::SMALLC) {
        const int col = l31 < C::CIN * 9 ? l31 : 0;                // col = tap*CIN + c
        const int tap = col / C::CIN, c = col - tap * C::CIN;
        boff = c * C::PLANE_X + (tap / 3) * C::PW + (tap % 3) + 2 * half;
    } else {
        boff = (cb * 32 + l31) * C::PLANE_X + ky * C::PW + 2 * half;  // + u*UNIT_X + 2*oyl*PW + 2*ox + kx
    }
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    lds_zero<NT>(lds, C::LDS_FLOATS, tid);

#pragma unroll 1
    for (int unit0 = blockIdx.x * C::NU; unit0 < total_units; unit0 += G * C::NU) {
        __syncthreads();
        // ---- stage x bands and gy bands (pads were zeroed once; data cells are always rewritten) ----
#pragma unroll 1
        for (int u = 0; u < C::NU; ++u) {
            const int unit = unit0 + u;
            const bool uvalid = unit < total_units;
            const int b = uvalid ? unit / C::NB : 0, band = unit % C::NB;
            const int bx = bidx ? bidx[b] : b;         // optional batch gather for the first layer's input
            const void* img = C::U8 ? (const void*)((const uint8_t*)xin + (size_t)bx * bstride)
                                    : (const void*)((const float*)xin + (size_t)bx * bstride);
            stage_x_band<C::CIN, C::H, C::W, C::IR, C::PW, C::PLANE_X, C::U8, NT>(xs + u * C::UNIT_X, img,
                                                                                  2 * band * C::R - 1, uvalid, tid);
            stage_y_band<C::COUT, C::HO, C::WO, C::R, C::POW, C::PLANE_Y, NT>(
                ys + u * C::UNIT_Y, gy + (size_t)b * C::COUT * C::HO * C::WO, band * C::R, uvalid, tid);
        }
        __syncthreads();
        // ---- K loop over the pixels of the staged units ----
#pragma unroll 1
        for (int u = 0; u < C::NU; ++u) {
            if constexpr (C::SMALLC) {
#pragma unroll 1
                for (int oyl = wave; oyl < C::R; oyl += C::NW) {
                    const int ao = aoff + u * C::UNIT_Y + oyl * C::POW;
                    const int bo = boff + u * C::UNIT_X + 2 * oyl * C::PW;
#pragma unroll 3
                    for (int s = 0; s < C::HSTEPS; ++s) {
                        const float a = ys[ao + 2 * s];
                        const float bv = xs[bo + 4 * s];
                        bsum += a;
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[0], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll 1
                for (int oyl = 0; oyl < C::R; ++oyl) {
                    const int ao = aoff + u * C::UNIT_Y + oyl * C::POW;
                    const int bo = boff + u * C::UNIT_X + 2 * oyl * C::PW;
#pragma unroll 3
                    for (int s = 0; s < C::HSTEPS; ++s) {
                        const float a = ys[ao + 2 * s];
                        const float b0 = xs[bo + 4 * s];
                        const float b1 = xs[bo + 4 * s + 1];
                        const float b2 = xs[bo + 4 * s + 2];
                        bsum += a;
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- write this workgroup's partial slab ----
    float* slab = slabs + (size_t)blockIdx.x * C::SLAB;
    bsum += __shfl_down(bsum, 32, 64);
    if constexpr (C::SMALLC) {
        // cross-wave (K-split) reduction through LDS, fixed order
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
            lds[wave * 1024 + n * 32 + l31] = acc[0][r];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += NT)
            slab[e] = (lds[e] + lds[1024 + e]) + (lds[2048 + e] + lds[3072 + e]);
        __syncthreads();
        if (half == 0) lds[wave * 32 + l31] = bsum;
        __syncthreads();
        if (tid < 32) slab[C::COUT * 32 + tid] = (lds[tid] + lds[32 + tid]) + (lds[64 + tid] + lds[96 + tid]);
    } else {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[(n * 9 + ky * 3 + kx) * C::CIN + cb * 32 + l31] = acc[kx][r];
            }
        }
        if (cb == 0 && ky == 0 && half == 0) slab[C::COUT * 9 * C::CIN + nb * 32 + l31] = bsum;
    }
}

// ------------------------------------------------------------------------------------------
// slab reduction -> OIHW gradient arena
// ------------------------------------------------------------------------------------------
struct RedSeg { int slab_off; int slab_sz; int G; int cin; int cout; int smallc; int gw; int gb; };
struct RedTable { RedSeg seg[5]; int start[6]; };

__global__ void __launch_bounds__(256)
img_wgrad_reduce_kernel(RedTable T, const float* __restrict__ slabs, float* __restrict__ grads) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T.start[5]) return;
    int l = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) if (j >= T.start[i]) l = i;
    const RedSeg S = T.seg[l];
    const int e = j - T.start[l];
    const float* p = slabs + S.slab_off + e;
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < S.G; ++g) s += p[(size_t)g * S.slab_sz];
    const int nw = S.smallc ? S.cout * 32 : S.cout * 9 * S.cin;
    if (e >= nw) { grads[S.gb + (e - nw)] = s; return; }
    if (S.smallc) {
        const int n = e / 32, col = e % 32;
        if (col < S.cin * 9) {
            const int tap = col / S.cin, c = col - tap * S.cin;
            grads[S.gw + (n * S.cin + c) * 9 + tap] = s;
        }
    } else {
        const int c = e % S.cin, tap = (e / S.cin) % 9, n = e / (9 * S.cin);
        grads[S.gw + (n * S.cin + c) * 9 + tap] = s;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <class C>
static int launch_dgrad(var_ctx* c, hipStream_t s, const float* gy, const float* wd, const float* x, float* gx, int B,
                        int layer) {
    ProfScope prof(c, s, TAG_IMG_DGRAD0 + layer);
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_dgrad_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const int units = B * C::NB;
    hipLaunchKernelGGL(img_dgrad_kernel<C>, dim3((units + C::NU - 1) / C::NU), dim3(C::NW * 64), C::LDS_BYTES, s,
                       gy, wd, x, gx, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

template <class C>
static int launch_wgrad(var_ctx* c, hipStream_t s, const void* x, long bstride, const float* gy, float* slabs,
                        int B, int G, int layer) {
    const int* bidx = layer == 0 ? c->saved_index : nullptr;
    ProfScope prof(c, s, TAG_IMG_WGRAD0 + layer);
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_wgrad_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(img_wgrad_kernel<C>, dim3(G), dim3(C::NW * 64), C::LDS_BYTES, s, x, bstride, bidx, gy, slabs, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

//                 CIN COUT  H  RI NU NW
using D84_1 = DgCfg<32, 32, 42, 6, 2, 4>;
using D84_2 = DgCfg<32, 64, 21, 22, 1, 4>;
using D84_3 = DgCfg<64, 64, 11, 12, 2, 6>;
using D84_4 = DgCfg<64, 64, 6, 6, 7, 4>;
using D96_1 = DgCfg<32, 32, 48, 8, 1, 3>;
using D96_2 = DgCfg<32, 64, 24, 8, 2, 3>;
using D96_3 = DgCfg<64, 64, 12, 12, 2, 6>;
using D96_4 = DgCfg<64, 64, 6, 6, 7, 4>;

//                 CIN COUT  H   U8    R  NU
using W84_0u = WgCfg<3, 32, 84, true, 6, 1>;
using W84_0f = WgCfg<3, 32, 84, false, 6, 1>;
using W84_1 = WgCfg<32, 32, 42, false, 3, 1>;
using W84_2 = WgCfg<32, 64, 21, false, 6, 1>;
using W84_3 = WgCfg<64, 64, 11, false, 6, 1>;
using W84_4 = WgCfg<64, 64, 6, false, 3, 4>;
using W96_0u = WgCfg<3, 32, 96, true, 6, 1>;
using W96_0f = WgCfg<3, 32, 96, false, 6, 1>;
using W96_1 = WgCfg<32, 32, 48, false, 3, 1>;
using W96_2 = WgCfg<32, 64, 24, false, 6, 1>;
using W96_3 = WgCfg<64, 64, 12, false, 6, 1>;
using W96_4 = WgCfg<64, 64, 6, false, 3, 4>;

template <class C>
static int wg_groups(int B, int gmax) {
    const int need = (B * C::NB + C::NU - 1) / C::NU;
    return need < gmax ? need : gmax;
}

// split-K workgroup counts per layer (also sizes the slab workspace in var_plan)
static const int kWgG[5] = {256, 256, 128, 64, 32};
static const int kSlabSz[5] = {32 * 32 + 32, 32 * 9 * 32 + 32, 64 * 9 * 32 + 64, 64 * 9 * 64 + 64, 64 * 9 * 64 + 64};

size_t img_slab_floats() {
    size_t t = 0;
    for (int i = 0; i < 5; i++) t += (size_t)kWgG[i] * kSlabSz[i];
    return t;
}

int launch_img_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    int rc;
    size_t so[5];
    size_t o = 0;
    for (int i = 0; i < 5; i++) { so[i] = o; o += (size_t)kWgG[i] * kSlabSz[i]; }
    const int H = c->H;
    int G[5] = {0, 0, 0, 0, 0};
    const long bs[6] = {c->saved_bstride, 32L * c->hs[1] * c->hs[1], 32L * c->hs[2] * c->hs[2],
                        64L * c->hs[3] * c->hs[3], 64L * c->hs[4] * c->hs[4], 0};
#define DG(CFG, l) do { if ((rc = launch_dgrad<CFG>(c, s, c->gact[l + 1], c->wpack + K.img_d[l], c->act[l], c->gact[l], B, l)) != VAR_OK) return rc; } while (0)
#define WG(CFG, l, X) do { G[l] = wg_groups<CFG>(B, kWgG[l]); if ((rc = launch_wgrad<CFG>(c, s, X, bs[l], c->gact[l + 1], c->slabs + so[l], B, G[l], l)) != VAR_OK) return rc; } while (0)
    if (H == 84) {
        WG(W84_4, 4, c->act[4]); DG(D84_4, 4);
        WG(W84_3, 3, c->act[3]); DG(D84_3, 3);
        WG(W84_2, 2, c->act[2]); DG(D84_2, 2);
        WG(W84_1, 1, c->act[1]); DG(D84_1, 1);
        if (c->saved_u8) { WG(W84_0u, 0, c->saved_image); } else { WG(W84_0f, 0, c->saved_image); }
    } else if (H == 96) {
        WG(W96_4, 4, c->act[4]); DG(D96_4, 4);
        WG(W96_3, 3, c->act[3]); DG(D96_3, 3);
        WG(W96_2, 2, c->act[2]); DG(D96_2, 2);
        WG(W96_1, 1, c->act[1]); DG(D96_1, 1);
        if (c->saved_u8) { WG(W96_0u, 0, c->saved_image); } else { WG(W96_0f, 0, c->saved_image); }
    } else {
        VAR_SET_ERR(c, "unsupported image size %d (84 or 96)", H);
        return VAR_ERR_ARG;
    }
#undef DG
#undef WG
    RedTable T{};
    int st = 0;
    for (int i = 0; i < 5; i++) {
        T.seg[i] = RedSeg{(int)so[i], kSlabSz[i], G[i], kImgCh[i], kImgCh[i + 1], i == 0 ? 1 : 0,
                          L.img_w[i], L.img_b[i]};
        T.start[i] = st;
        st += kSlabSz[i];
    }
    T.start[5] = st;
    ProfScope prof(c, s, TAG_IMG_WREDUCE);
    hipLaunchKernelGGL(img_wgrad_reduce_kernel, dim3((st + 255) / 256), dim3(256), 0, s, T, c->slabs, grads);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
