// Image CNN weight gradients (autograd of models/pretext/arm_pretext_model.py:9-18):
//   dW[n][c][tap] = sum_{b,oy,ox} gy[b][n][oy][ox] * x[b][c][2oy+ky-1][2ox+kx-1],  db[n] = sum gy[b][n][..]
// on the gfx950 f32 matrix cores:  D[n][c] per tap, reduction index = output pixels.
//
// Work decomposition
//   blockIdx.y = (32-channel block of n, 32-channel block of c); a workgroup has 3 waves, wave = ky,
//   each wave owns the three kx tiles of its ky (A operand shared by the three MFMAs);
//   blockIdx.x strides over "units" (R output rows of one image): split-K across workgroups.
//   The first layer (CIN = 3) packs (tap, c) into one 32-wide tile and splits K over 4 waves instead.
// Software pipeline (the point of this kernel)
//   every lane issues ALL global loads of the NEXT unit into registers right after the barrier
//   that precedes the MFMAs of the CURRENT unit, and writes them to LDS after those MFMAs:
//   HBM/L2 latency is overlapped with matrix-core work instead of being paid per batch of loads.
// Determinism
//   each workgroup writes one partial slab; img_wgrad_reduce_kernel sums slabs in a fixed order.
#include "var_common.h"

template <int CIN_, int COUT_, int H_, bool U8_, int R_, int NU_>
struct WgCfg {
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
    static constexpr int NW = SMALLC ? 4 : 3;
    static constexpr int NT = NW * 64;
    static constexpr int XS = (NU * UNIT_X + 3) & ~3, YS = NU * UNIT_Y;
    static constexpr int LDS_FLOATS = (((XS + YS) > (SMALLC ? 4096 : 0) ? (XS + YS) : 4096) + 3) & ~3;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int SLAB = SMALLC ? (32 * 32 + 32) : (32 * 9 * 32 + 32);
    static constexpr int HSTEPS = (WO + 1) / 2;
    // staging vectors
    static constexpr int VX = U8 ? 4 : (W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1));
    static constexpr int WVX = W / VX;
    static constexpr int NXV = XC * IR * WVX;                         // per unit
    static constexpr int XI = (NU * NXV + NT - 1) / NT;
    static constexpr int VY = (WO % 4 == 0) ? 4 : (WO % 2 == 0 ? 2 : 1);
    static constexpr int WVY = WO / VY;
    static constexpr int NYV = 32 * R * WVY;
    static constexpr int YI = (NU * NYV + NT - 1) / NT;
    static constexpr int XREG = U8 ? 1 : VX;                          // registers per staged x vector
};

template <class C>
__global__ void __launch_bounds__(C::NW * 64)
img_wgrad_kernel(const void* __restrict__ xin, long bstride, const int* __restrict__ bidx,
                 const float* __restrict__ gy, float* __restrict__ slabs, int B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ys = lds + C::XS;
    constexpr int NT = C::NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int G = gridDim.x;
    const int combo = blockIdx.y;
    const int nb = combo / C::CBLK, cb = combo - nb * C::CBLK;

    // ---- per-thread staging descriptors (fixed for the whole kernel): u | c | r | xv packed ----
    int xpk[C::XI], ypk[C::YI];
#pragma unroll
    for (int i = 0; i < C::XI; ++i) {
        const int e = tid + i * NT;
        if (e < C::NU * C::NXV) {
            const int u = e / C::NXV, rem = e - u * C::NXV;
            const int c = rem / (C::IR * C::WVX), rem2 = rem - c * (C::IR * C::WVX);
            const int r = rem2 / C::WVX, xv = rem2 - r * C::WVX;
            xpk[i] = xv | (r << 8) | (c << 16) | (u << 24);
        } else {
            xpk[i] = -1;
        }
    }
#pragma unroll
    for (int i = 0; i < C::YI; ++i) {
        const int e = tid + i * NT;
        if (e < C::NU * C::NYV) {
            const int u = e / C::NYV, rem = e - u * C::NYV;
            const int n = rem / (C::R * C::WVY), rem2 = rem - n * (C::R * C::WVY);
            const int r = rem2 / C::WVY, xv = rem2 - r * C::WVY;
            ypk[i] = xv | (r << 8) | (n << 16) | (u << 24);
        } else {
            ypk[i] = -1;
        }
    }
    float xr[C::XI][C::XREG];
    float yr[C::YI][C::VY];

    // NOTE: the loaded values are not touched here (no select on them): any VALU use would make the
    // compiler wait for the loads before the MFMAs they are meant to overlap with.  Rows that fall
    // outside the image are loaded from a clamped address and zeroed in store_stage.
    auto issue_loads = [&](int unit0) {
        // image row of each staged unit, resolved ONCE (an index load inside the element loop would put
        // a vmcnt(0) wait between every pair of data loads)
        int ub[C::NU], uband[C::NU];
#pragma unroll
        for (int u = 0; u < C::NU; ++u) {
            const int unit = unit0 + u;
            const bool uv = unit < total_units;
            const int bo = uv ? unit / C::NB : 0;
            uband[u] = uv ? unit - bo * C::NB : 0;
            ub[u] = bo;
        }
        int uimg[C::NU];
#pragma unroll
        for (int u = 0; u < C::NU; ++u) uimg[u] = bidx ? bidx[ub[u]] : ub[u];
#pragma unroll
        for (int i = 0; i < C::XI; ++i) {
            const int pk = xpk[i], pkk = pk < 0 ? 0 : pk;      // lanes without an element read element 0
            const int xv = pkk & 0xff, r = (pkk >> 8) & 0xff, c = (pkk >> 16) & 0xff, u = (pkk >> 24) & 0x7f;
            const int unit = unit0 + u;
            const bool uvalid = pk >= 0 && unit < total_units;
            int b = uimg[0], band = uband[0];
#pragma unroll
            for (int q = 1; q < C::NU; ++q) if (u == q) { b = uimg[q]; band = uband[q]; }
            const int iy = 2 * band * C::R - 1 + r;
            const bool ok = uvalid && iy >= 0 && iy < C::H;
            const int iyc = ok ? iy : 0;
            const size_t so = (size_t)b * bstride + ((cb * C::XC + c) * C::H + iyc) * C::W + xv * C::VX;
            if constexpr (C::U8) {
                const uint32_t q = *(const uint32_t*)((const uint8_t*)xin + so);
                xr[i][0] = __uint_as_float(q);
            } else if constexpr (C::VX == 4) {
                const float4 q = *(const float4*)((const float*)xin + so);
                xr[i][0] = q.x; xr[i][1] = q.y; xr[i][2] = q.z; xr[i][3] = q.w;
            } else if constexpr (C::VX == 2) {
                const float2 q = *(const float2*)((const float*)xin + so);
                xr[i][0] = q.x; xr[i][1] = q.y;
            } else {
                xr[i][0] = ((const float*)xin)[so];
            }
        }
#pragma unroll
        for (int i = 0; i < C::YI; ++i) {
            const int pk = ypk[i], pkk = pk < 0 ? 0 : pk;
            const int xv = pkk & 0xff, r = (pkk >> 8) & 0xff, n = (pkk >> 16) & 0xff, u = (pkk >> 24) & 0x7f;
            const int unit = unit0 + u;
            const bool uvalid = pk >= 0 && unit < total_units;
            int b = ub[0], band = uband[0];
#pragma unroll
            for (int q = 1; q < C::NU; ++q) if (u == q) { b = ub[q]; band = uband[q]; }
            const int oy = band * C::R + r;
            const bool ok = uvalid && oy < C::HO;
            const int oyc = ok ? oy : 0;
            const float* p = gy + ((size_t)(b * C::COUT + nb * 32 + n) * C::HO + oyc) * C::WO + xv * C::VY;
            if constexpr (C::VY == 4) {
                const float4 q = *(const float4*)p;
                yr[i][0] = q.x; yr[i][1] = q.y; yr[i][2] = q.z; yr[i][3] = q.w;
            } else if constexpr (C::VY == 2) {
                const float2 q = *(const float2*)p;
                yr[i][0] = q.x; yr[i][1] = q.y;
            } else {
                yr[i][0] = *p;
            }
        }
    };
    auto store_stage = [&](int unit0) {
#pragma unroll
        for (int i = 0; i < C::XI; ++i) {
            const int pk = xpk[i];
            if (pk >= 0) {
                const int xv = pk & 0xff, r = (pk >> 8) & 0xff, c = (pk >> 16) & 0xff, u = (pk >> 24) & 0x7f;
                const int unit = unit0 + u;
                const int band = unit % C::NB;
                const int iy = 2 * band * C::R - 1 + r;
                const bool ok = unit < total_units && iy >= 0 && iy < C::H;
                float* d = xs + u * C::UNIT_X + c * C::PLANE_X + r * C::PW + 1 + xv * C::VX;
                if constexpr (C::U8) {
                    const uint32_t q = ok ? __float_as_uint(xr[i][0]) : 0u;
                    d[0] = (float)(q & 0xff) / 255.f; d[1] = (float)((q >> 8) & 0xff) / 255.f;
                    d[2] = (float)((q >> 16) & 0xff) / 255.f; d[3] = (float)(q >> 24) / 255.f;
                } else {
#pragma unroll
                    for (int j = 0; j < C::VX; ++j) d[j] = ok ? xr[i][j] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < C::YI; ++i) {
            const int pk = ypk[i];
            if (pk >= 0) {
                const int xv = pk & 0xff, r = (pk >> 8) & 0xff, n = (pk >> 16) & 0xff, u = (pk >> 24) & 0x7f;
                const int unit = unit0 + u;
                const bool ok = unit < total_units && (unit % C::NB) * C::R + r < C::HO;
                float* d = ys + u * C::UNIT_Y + n * C::PLANE_Y + r * C::POW + xv * C::VY;
#pragma unroll
                for (int j = 0; j < C::VY; ++j) d[j] = ok ? yr[i][j] : 0.f;
            }
        }
    };

    // ---- lane offsets for the MFMA operands ----
    const int ky = C::SMALLC ? 0 : wave;
    const int aoff = l31 * C::PLANE_Y + half;                       // + u*UNIT_Y + oyl*POW + 2s
    int boff;
    if constexpr (C::SMALLC) {
        const int col = l31 < C::CIN * 9 ? l31 : 0;                  // col = tap*CIN + c
        const int tap = col / C::CIN, c = col - tap * C::CIN;
        boff = c * C::PLANE_X + (tap / 3) * C::PW + (tap % 3) + 2 * half;
    } else {
        boff = l31 * C::PLANE_X + ky * C::PW + 2 * half;             // + u*UNIT_X + 2*oyl*PW + 4s + kx
    }
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // zero the pads once (data cells are rewritten for every unit)
    {
        float4 z = {0.f, 0.f, 0.f, 0.f};
        for (int e = tid; e < C::LDS_FLOATS / 4; e += NT) ((float4*)lds)[e] = z;
    }
    const int first = blockIdx.x * C::NU;
    if (first < total_units) issue_loads(first);
#pragma unroll 1
    for (int unit0 = first; unit0 < total_units; unit0 += G * C::NU) {
        __syncthreads();                              // previous unit's MFMAs are done reading LDS
        store_stage(unit0);
        __syncthreads();
        if (unit0 + G * C::NU < total_units) issue_loads(unit0 + G * C::NU);   // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int u = 0; u < C::NU; ++u) {
            if constexpr (C::SMALLC) {
                // K-split over the 4 waves: items = (row, half row)
#pragma unroll 1
                for (int it = wave; it < 2 * C::R; it += C::NW) {
                    const int oyl = it >> 1, s0 = (it & 1) ? (C::HSTEPS + 1) / 2 : 0;
                    const int s1 = (it & 1) ? C::HSTEPS : (C::HSTEPS + 1) / 2;
                    const int ao = aoff + u * C::UNIT_Y + oyl * C::POW;
                    const int bo = boff + u * C::UNIT_X + 2 * oyl * C::PW;
#pragma unroll 4
                    for (int s = s0; s < s1; ++s) {
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
#pragma unroll
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
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- write this workgroup's partial slab ----
    float* slab = slabs + ((size_t)blockIdx.x * C::NCOMBO + combo) * C::SLAB;
    bsum += __shfl_down(bsum, 32, 64);
    if constexpr (C::SMALLC) {
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
        if (tid < 32) slab[1024 + tid] = (lds[tid] + lds[32 + tid]) + (lds[64 + tid] + lds[96 + tid]);
    } else {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[(n * 9 + ky * 3 + kx) * 32 + l31] = acc[kx][r];
            }
        }
        if (ky == 0 && half == 0) slab[32 * 9 * 32 + l31] = bsum;
    }
}

// ------------------------------------------------------------------------------------------
// slab reduction -> OIHW gradient arena
// ------------------------------------------------------------------------------------------
struct RedSeg { int slab_off; int slab_sz; int G; int ncombo; int cblk; int cin; int smallc; int gw; int gb; };
struct RedTable { RedSeg seg[5]; int start[6]; };

__global__ void __launch_bounds__(256)
img_wgrad_reduce_kernel(RedTable T, const float* __restrict__ slabs, float* __restrict__ grads) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T.start[5]) return;
    int l = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) if (j >= T.start[i]) l = i;
    const RedSeg S = T.seg[l];
    const int e = j - T.start[l];                    // element of the (combo, slab) space of this layer
    const float* p = slabs + S.slab_off + e;
    const size_t gstride = (size_t)S.ncombo * S.slab_sz;
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < S.G; ++g) s += p[(size_t)g * gstride];
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
//                   CIN COUT  H   U8    R  NU
using W84_0u = WgCfg<3, 32, 84, true, 6, 1>;
using W84_0f = WgCfg<3, 32, 84, false, 6, 1>;
using W84_1 = WgCfg<32, 32, 42, false, 3, 1>;
using W84_2 = WgCfg<32, 64, 21, false, 3, 1>;
using W84_3 = WgCfg<64, 64, 11, false, 6, 1>;
using W84_4 = WgCfg<64, 64, 6, false, 3, 4>;
using W96_0u = WgCfg<3, 32, 96, true, 6, 1>;
using W96_0f = WgCfg<3, 32, 96, false, 6, 1>;
using W96_1 = WgCfg<32, 32, 48, false, 3, 1>;
using W96_2 = WgCfg<32, 64, 24, false, 3, 1>;
using W96_3 = WgCfg<64, 64, 12, false, 6, 1>;
using W96_4 = WgCfg<64, 64, 6, false, 3, 4>;

// split-K workgroups (grid.x) per layer; grid.y = channel-block combos.  Also sizes the slab workspace.
static const int kWgG[5] = {256, 256, 128, 64, 64};
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

template <class C>
static int launch_wgrad(var_ctx* c, hipStream_t s, const void* x, long bstride, const float* gy, int B, int layer) {
    static_assert(C::NCOMBO * C::SLAB <= 4 * 9248, "slab sizing");
    const int* bidx = layer == 0 ? c->saved_index : nullptr;
    ProfScope prof(c, s, TAG_IMG_WGRAD0 + layer);
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_wgrad_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const int need = (B * C::NB + C::NU - 1) / C::NU;
    const int G = need < kWgG[layer] ? need : kWgG[layer];
    c->wg_groups[layer] = G;
    hipLaunchKernelGGL(img_wgrad_kernel<C>, dim3(G, C::NCOMBO), dim3(C::NT), C::LDS_BYTES, s, x, bstride, bidx, gy,
                       c->slabs + slab_offset(layer), B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// weight gradient of image conv `layer` (0..4): x = that layer's input, gy = gact[layer+1]
int launch_img_wgrad(var_ctx* c, hipStream_t s, int layer, const void* x, long bstride, int is_u8, const float* gy, int B) {
    const int H = c->H;
#define W2(A, Bc) (H == 84 ? launch_wgrad<A>(c, s, x, bstride, gy, B, layer) : launch_wgrad<Bc>(c, s, x, bstride, gy, B, layer))
    switch (layer) {
        case 0: return is_u8 ? W2(W84_0u, W96_0u) : W2(W84_0f, W96_0f);
        case 1: return W2(W84_1, W96_1);
        case 2: return W2(W84_2, W96_2);
        case 3: return W2(W84_3, W96_3);
        default: return W2(W84_4, W96_4);
    }
#undef W2
}

int launch_img_wgrad_reduce(var_ctx* c, hipStream_t s, float* grads) {
    const ParamLayout& L = c->pl;
    RedTable T{};
    int st = 0;
    for (int i = 0; i < 5; i++) {
        T.seg[i] = RedSeg{(int)slab_offset(i), kSlabSz[i], c->wg_groups[i], kCombo[i], i < 3 ? 1 : 2, kImgCh[i],
                          i == 0 ? 1 : 0, L.img_w[i], L.img_b[i]};
        T.start[i] = st;
        st += kCombo[i] * kSlabSz[i];
    }
    T.start[5] = st;
    ProfScope prof(c, s, TAG_IMG_WREDUCE);
    hipLaunchKernelGGL(img_wgrad_reduce_kernel, dim3((st + 255) / 256), dim3(256), 0, s, T, c->slabs, grads);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
