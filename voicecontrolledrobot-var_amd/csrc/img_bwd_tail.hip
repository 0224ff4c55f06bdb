// Tail of the image CNN backward in ONE kernel: the data gradient of the second conv and the weight
// gradient of the first conv (autograd of models/pretext/arm_pretext_model.py:9-12 under
// loss.backward(), VAR/pretext_VAR.py:68).
//
// Unfused, the gradient of the first activation map gact1 (B x 32 x H x H floats -- with the first
// activation itself the largest tensor of the step) is written to HBM by the dgrad kernel and read
// back once by the first layer's wgrad kernel, which has almost no arithmetic (K = 27).  Here a
// persistent workgroup keeps each band of gact1 in LDS instead:
//   per tile (RI rows of one image):
//     1. stage the gy band (gact2) and the u8/f32 input-image band into LDS
//     2. dgrad on the matrix cores, D[c][pixel pair] per row-parity class, 3 equal K parts
//        {py=0 | py=1 row tap 0 | py=1 row tap 1} x NPB pixel blocks = NW waves (img_conv_bwd.hip)
//     3. ReLU mask (bit pattern left by the first conv's forward, c->relu1) and write the tile to LDS as gxs[c][row][col]; the second py=1 part adds
//        its half after a barrier (fixed order)
//     4. wgrad of conv1 on the matrix cores, D[n][(tap,c)] += gxs[n][pixel] * image[c][2oy+ky-1][2ox+kx-1],
//        the tile's pixels split over the NW waves; accumulators stay in registers across tiles
//   at the end the waves are folded through LDS in a fixed order into one slab per workgroup
//   (img_wgrad_reduce_kernel sums the slabs -> deterministic, no float atomics).
// gact1 never exists in HBM.
#include <stdlib.h>

#include <type_traits>

#include "img_stage.h"

template <int H_, bool U8_, int RI_, int NU_>
struct TailCfg {
    static constexpr int H = H_, W = H_, RI = RI_, NU = NU_; // gx = act1 plane (42 / 48); NU bands per tile
    static constexpr bool U8 = U8_;
    static constexpr int CH = 32;                            // channels of gx and of gy
    static constexpr int HI = 2 * H;                         // input image (84 / 96)
    static constexpr int HO = H / 2, WO = HO;                // gy = gact2 plane (21 / 24)
    static constexpr int NB = H / RI;                        // bands per image
    static constexpr int NR = RI / 2 + 1;                    // gy rows per band
    static constexpr int POW = WO + 1;                       // + zero column at ox = WO
    static constexpr int PLANE_Y = NR * POW;
    static constexpr int UNIT_Y = CH * PLANE_Y;
    static constexpr int WH = W / 2;                         // pixel pairs per row
    static constexpr int PPU = (RI / 2) * WH;                // pixel pairs per band and row-parity class
    static constexpr int NPB = (NU * PPU + 31) / 32;
    static constexpr int NW = 3 * NPB, NT = NW * 64;
    static constexpr int PLANE_G = RI * W + 2;               // even (8-byte LDS stores of pixel pairs), = 30 mod 32 for 84:
                                                             // the wgrad A reads (lanes differ in channel) are 2-way conflicts at worst
    static constexpr int UNIT_G = CH * PLANE_G;
    static constexpr int IRX = 2 * RI + 1;                   // image rows per band
    static constexpr int PWX = HI + 3;
    static constexpr int PLANE_X = (IRX * PWX) | 1;
    static constexpr int UNIT_X = 3 * PLANE_X;
    static constexpr int GYS = 0;
    static constexpr int GXS = (NU * UNIT_Y + 3) & ~3;
    static constexpr int IMS = (GXS + NU * UNIT_G + 3) & ~3;
    static constexpr int WDS = (IMS + NU * UNIT_X + 3) & ~3; // dgrad filter, 9 taps x CH x CH
    static constexpr int LUT = WDS + 9 * CH * CH;            // x / 255.f, x = 0..255
    static constexpr int ZPAD = LUT + 256;
    static constexpr int LDS_FLOATS = (ZPAD + 4) > NW * 1024 ? (ZPAD + 4) : NW * 1024;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static constexpr int HSTEPS = W / 2;                     // wgrad k-steps per row (2 pixels each)
    static constexpr int SLAB = 32 * 32 + 32;
    static_assert(H % RI == 0 && RI % 2 == 0 && W % 2 == 0, "whole bands of row pairs, even width");
    static_assert(NU * RI <= NW, "the wgrad phase gives one gx row to a wave");
    static_assert(NU <= 2, "at most two bands per tile");
};

// (a __device__ body so that it can also run as one half of a fused launch, img_conv_bwd.hip: bx / G stand for
// blockIdx.x / gridDim.x of a stand-alone launch)
template <class C>
__device__ __forceinline__ void img_bwd_tail_body(const float* __restrict__ gy, const float* __restrict__ wd,
                                                  const uint16_t* __restrict__ relu_bits, const void* __restrict__ image,
                                                  long bstride, const int* __restrict__ bidx, float* __restrict__ slabs,
                                                  int B, int bx, int G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NT;
    using XT = typename std::conditional<C::U8, uint8_t, float>::type;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int ntiles = (total_units + C::NU - 1) / C::NU;
    float* gys = lds + C::GYS;
    float* ims = lds + C::IMS;

    // ---- dgrad lane constants (K-split path of img_dgrad_kernel, one item per wave) ----
    const int wv = wave % C::NPB, part = wave / C::NPB;
    const int py = part ? 1 : 0, ky = part == 0 ? 1 : (part == 1 ? 0 : 2), doy = part == 1 ? 1 : 0;
    int pp = wv * 32 + l31;
    const bool ppvalid = pp < C::NU * C::PPU;
    if (!ppvalid) pp = 0;
    const int pu = pp / C::PPU, pq = pp - pu * C::PPU;                 // band of the tile, pair inside it
    const int pj = pq / C::WH, pi = pq - pj * C::WH;
    const int rl = 2 * pj + py;                                       // gx row inside the band
    const int lb = pu * C::UNIT_Y + pj * C::POW + pi + half * C::PLANE_Y + doy * C::POW;
    const int wco = half * C::CH + l31 + (ky * 3) * C::CH * C::CH;      // filter element: + (kx*CH + n)*CH
    const int gxo = rl * C::W + 2 * pi;                                // inside a plane of the band
    const int gxl = C::GXS + pu * C::UNIT_G + gxo;                     // + c*PLANE_G

    // ---- wgrad lane constants (first-layer form of img_wgrad_kernel: columns = (tap, c)) ----
    int boff;
    {
        const int col = l31 < 27 ? l31 : 0;
        const int tap = col / 3, c = col - tap * 3;
        boff = C::IMS + c * C::PLANE_X + (tap / 3) * C::PWX + (tap % 3) + 2 * half;
    }
    // one gx row of the tile per wave (waves >= NU*RI idle in this phase): operand offsets are base + immediates
    const bool wlive = wave < C::NU * C::RI;
    const int wu = wlive ? wave / C::RI : 0, wr = wlive ? wave - wu * C::RI : 0;
    const int abase = wlive ? C::GXS + wu * C::UNIT_G + l31 * C::PLANE_G + wr * C::W + half : C::ZPAD;
    const int astep = wlive ? 2 : 0;
    const int bbase = boff + wu * C::UNIT_X + 2 * wr * C::PWX;
    f32x16 wacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
    // The dgrad filter lives in LDS for the whole (persistent) kernel: no filter traffic and no filter
    // registers inside the tile loop.
    constexpr int NP = C::CH / 2;
    for (int e = tid; e < 9 * C::CH * C::CH / 4; e += NT) ((float4*)(lds + C::WDS))[e] = ((const float4*)wd)[e];
    if (tid < 256) lds[C::LUT + tid] = (float)tid / 255.f;
    const float* wl = lds + C::WDS + wco;
    const float* lut = lds + C::LUT;
    float bsum = 0.f;

    // pad cells, once: gy column ox = WO, image columns 0 and > HI, the always-zero A cell
    lds_zero_cols<NT>(gys, C::NU * C::CH * C::NR, C::POW, C::WO, 1, tid);
#pragma unroll
    for (int c = 0; c < 3 * C::NU; ++c) {
        float* pl = ims + (c / 3) * C::UNIT_X + (c % 3) * C::PLANE_X;
        lds_zero_cols<NT>(pl, C::IRX, C::PWX, 0, 1, tid);
        lds_zero_cols<NT>(pl, C::IRX, C::PWX, C::HI + 1, C::PWX - C::HI - 1, tid);
    }
    if (tid < 4) lds[C::ZPAD + tid] = 0.f;

    BandCopy<C::CH, C::HO, C::WO, C::NR, false, NT, C::NU> cy;
    BandCopy<3, C::HI, C::HI, C::IRX, C::U8, NT, C::NU> cx;
    // wave-uniform bookkeeping of a tile: images and bands of its (up to two) units
    struct Tile { const float* gy0; const float* gy1; const XT* im0; const XT* im1; int ry0, ry1, rx0, rx1; bool ok0, ok1; };
    // (gi0, gi1: dataset rows of the two images -- the optional batch gather -- fetched one tile ahead by
    // gather_rows(), so that the scalar index load is never waited for in front of the image loads)
    auto gather_rows = [&](int tile, int& gi0, int& gi1) {
        const int u0 = tile * C::NU, u1 = u0 + C::NU - 1;
        const int b0 = u0 < total_units ? u0 / C::NB : 0, b1 = u1 < total_units ? u1 / C::NB : 0;
        gi0 = bidx ? bidx[b0] : b0;
        gi1 = bidx ? bidx[b1] : b1;
    };
    auto make_tile = [&](int tile, int gi0, int gi1) {
        Tile t;
        const int u0 = tile * C::NU, u1 = u0 + C::NU - 1;
        t.ok0 = u0 < total_units; t.ok1 = u1 < total_units;
        const int b0 = t.ok0 ? u0 / C::NB : 0, b1 = t.ok1 ? u1 / C::NB : 0;
        const int band0 = t.ok0 ? u0 - b0 * C::NB : 0, band1 = t.ok1 ? u1 - b1 * C::NB : 0;
        t.gy0 = gy + (size_t)b0 * C::CH * C::HO * C::WO; t.gy1 = gy + (size_t)b1 * C::CH * C::HO * C::WO;
        t.im0 = (const XT*)image + (size_t)gi0 * bstride;
        t.im1 = (const XT*)image + (size_t)gi1 * bstride;
        t.ry0 = band0 * (C::RI / 2); t.ry1 = band1 * (C::RI / 2);
        t.rx0 = 2 * band0 * C::RI - 1; t.rx1 = 2 * band1 * C::RI - 1;
        return t;
    };
    // ReLU bits of this lane's pixel pair (one u32 = two u16 written by the first conv's forward); 0 for a
    // band that does not exist, so that its gx rows become zeros
    auto load_mask = [&](int tile) -> uint32_t {
        const int unit = tile * C::NU + pu;
        if (unit >= total_units) return 0u;
        const int b = unit / C::NB, band = unit - b * C::NB;
        return *(const uint32_t*)(relu_bits + ((size_t)b * 2 + half) * C::H * C::W + (band * C::RI) * C::W + gxo);
    };

    int tile = bx;
    uint32_t m = 0;
    int gi0 = 0, gi1 = 0;
    if (tile < ntiles) {
        gather_rows(tile, gi0, gi1);
        const Tile t = make_tile(tile, gi0, gi1);
        cy.issue(t.gy0, t.gy1, t.ry0, t.ry1, tid);
        cx.issue(t.im0, t.im1, t.rx0, t.rx1, tid);
        m = load_mask(tile);
        cy.template store<C::UNIT_Y, C::PLANE_Y, C::POW, 0>(gys, lut, t.ry0, t.ry1, t.ok0, t.ok1, tid);
        cx.template store<C::UNIT_X, C::PLANE_X, C::PWX, 1>(ims, lut, t.rx0, t.rx1, t.ok0, t.ok1, tid);
    }
    if (tile + G < ntiles) gather_rows(tile + G, gi0, gi1);
    __syncthreads();
#pragma unroll 1
    for (; tile < ntiles; tile += G) {
        // ---- loads of the NEXT tile: in flight during this tile's matrix work ----
        const int next = tile + G;
        const bool more = next < ntiles;
        int tid_t = tid;                                   // opaque copy: see BandCopy
        asm volatile("" : "+v"(tid_t));
        uint32_t m_next = 0;
        const Tile tn = make_tile(more ? next : tile, gi0, gi1);
        if (next + G < ntiles) gather_rows(next + G, gi0, gi1);     // for the next iteration
        if (more) {
            cy.issue(tn.gy0, tn.gy1, tn.ry0, tn.ry1, tid_t);
            cx.issue(tn.im0, tn.im1, tn.rx0, tn.rx1, tid_t);
            m_next = load_mask(next);
        }

        // ---- dgrad: this wave's K part of its pixel block (filter slice already in registers) ----
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        {
            // gy operands are read from LDS one chunk (UC n-pairs) ahead of the MFMAs that use them; the
            // scheduling barriers keep hipcc from hoisting every LDS read of the loop to its top (spills)
            constexpr int UC = 4, NCH = NP / UC;
            float bb[2][UC][2], wa[2][UC][3];
            auto fetch = [&](int buf, int ch) {
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    bb[buf][u][0] = gys[lb + 2 * (ch * UC + u) * C::PLANE_Y];
                    bb[buf][u][1] = gys[lb + 2 * (ch * UC + u) * C::PLANE_Y + 1];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) wa[buf][u][kx] = wl[(kx * C::CH + 2 * (ch * UC + u)) * C::CH];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) fetch((ch + 1) & 1, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    // kx=1 -> px=0 (dox 0); kx=2 -> px=1 (dox 0); kx=0 -> px=1 (dox 1)
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ch & 1][u][1], bb[ch & 1][u][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ch & 1][u][2], bb[ch & 1][u][0], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ch & 1][u][0], bb[ch & 1][u][1], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- masked tile -> LDS (parts 0 and 1 write, part 2 adds afterwards: fixed order) ----
        // (bit r of the u16 -> all-ones / zero word by one signed bit-field extract, then AND)
        auto masked = [&](int r) {
            float2 g;
            g.x = __int_as_float(__float_as_int(acc0[r]) & ((int)(m << (31 - r)) >> 31));
            g.y = __int_as_float(__float_as_int(acc1[r]) & ((int)(m << (15 - r)) >> 31));
            return g;
        };
        if (ppvalid && part < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                *(float2*)(lds + gxl + c * C::PLANE_G) = masked(r);
            }
        }
        __syncthreads();                                   // gys is dead, gxs holds parts 0 and 1
        if (ppvalid && part == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                float2* d = (float2*)(lds + gxl + c * C::PLANE_G);
                const float2 g = masked(r), o = *d;
                *d = make_float2(o.x + g.x, o.y + g.y);
            }
        }
        __syncthreads();

        // next tile's gy band -> LDS (gys is dead since the barrier before last); these stores overlap the
        // matrix work of the other waves below
        if (more) cy.template store<C::UNIT_Y, C::PLANE_Y, C::POW, 0>(gys, lut, tn.ry0, tn.ry1, tn.ok0, tn.ok1, tid_t);
        // ---- wgrad of the first conv over this tile's pixels ----
        {
            constexpr int UC = 7, NCH = (C::HSTEPS + UC - 1) / UC;
            float wa[2][UC], wx[2][UC];
            auto fetch = [&](int buf, int ch) {
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int i = ch * UC + u < C::HSTEPS ? ch * UC + u : C::HSTEPS - 1;
                    wa[buf][u] = lds[abase + astep * i];
                    wx[buf][u] = lds[bbase + 4 * i];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) fetch((ch + 1) & 1, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    if (ch * UC + u < C::HSTEPS) {
                        bsum += wa[ch & 1][u];
                        wacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ch & 1][u], wx[ch & 1][u], wacc, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                   // ims and gxs are dead
        if (more) cx.template store<C::UNIT_X, C::PLANE_X, C::PWX, 1>(ims, lut, tn.rx0, tn.rx1, tn.ok0, tn.ok1, tid_t);
        m = m_next;
    }

    // ---- fold the waves through LDS (fixed order) and write this workgroup's partial slab ----
    float* slab = slabs + (size_t)bx * C::SLAB;
    bsum += __shfl_down(bsum, 32, 64);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
        lds[wave * 1024 + n * 32 + l31] = wacc[r];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += NT) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < C::NW; ++q) sum += lds[q * 1024 + e];
        slab[e] = sum;
    }
    __syncthreads();
    if (half == 0) lds[wave * 32 + l31] = bsum;
    __syncthreads();
    if (tid < 32) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < C::NW; ++q) sum += lds[q * 32 + tid];
        slab[1024 + tid] = sum;
    }
}

template <class C>
__global__ void __launch_bounds__(C::NT)
img_bwd_tail_kernel(const float* __restrict__ gy, const float* __restrict__ wd, const uint16_t* __restrict__ relu_bits,
                    const void* __restrict__ image, long bstride, const int* __restrict__ bidx,
                    float* __restrict__ slabs, int B) {
    img_bwd_tail_body<C>(gy, wd, relu_bits, image, bstride, bidx, slabs, B, blockIdx.x, gridDim.x);
}

//                    H    U8   RI NU
using T96u = TailCfg<48, true, 8, 1>;      // 1 band: 3 pixel blocks x 3 K parts = 9 waves
using T96f = TailCfg<48, false, 8, 1>;

template <class C>
static int launch_tail(var_ctx* c, hipStream_t s, int B) {
    ProfScope prof(c, s, TAG_IMG_DGRAD0 + 1);
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_bwd_tail_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    const int ntiles = (B * C::NB + C::NU - 1) / C::NU;
    const int G = ntiles < kTailG ? ntiles : kTailG;
    c->wg_groups[0] = G;
    hipLaunchKernelGGL(img_bwd_tail_kernel<C>, dim3(G), dim3(C::NT), C::LDS_BYTES, s, c->gact[2],
                       c->wpack + c->kl.img_d[1], c->relu1, c->saved_image, c->saved_bstride, c->saved_index,
                       c->slabs, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// dgrad of image conv 2 fused with the weight gradient of image conv 1; leaves layer 0's slabs
// (slab workspace offset 0, c->wg_groups[0] of them) for launch_img_wgrad_reduce
int launch_img_bwd_tail(var_ctx* c, hipStream_t s, int B) {
    return c->saved_u8 ? launch_tail<T96u>(c, s, B) : launch_tail<T96f>(c, s, B);
}
