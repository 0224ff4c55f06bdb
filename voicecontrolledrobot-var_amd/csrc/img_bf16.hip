// 3x3 stride-1 convolutions of the iTHOR image branch in the model's bf16 mode (models/pretext/ai2thor_pretext_model.py:
// 7-16; layers 2 and 3: 32 -> 32 at 96x96 and 32 -> 64 at 48x48, 1.7 of the step's 10 ms on the gather-GEMM), forward
// and data gradient as ONE kernel (the data gradient of a stride-1 3x3 convolution is the same correlation with the
// filters transposed and flipped), on v_mfma_f32_32x32x16_bf16.
//
// fp32 NCHW in and out (the pooling / masking kernels around these layers stay as they are): a tile's input patch is
// gathered once from the 8 channel planes of each pixel into a channel-innermost bf16 image in LDS ([plane of 8
// channels][row][W + 1 slots of 16 bytes]: the zero slot in front of a row is the left padding of that row and the right
// padding of the previous one), and every tap reads it as base + immediate offset (ds_read_b128, consecutive lanes =
// consecutive slots).  Tile = TR rows x W columns of one image; wave = MBW 32-pixel blocks x all output-channel blocks;
// two workgroups per CU, so one stages while the other multiplies.  Filters are re-packed per step into fragment order and
// streamed from L2 one filter row ahead.
#include "var_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ u32x4_t wload(__amdgpu_buffer_rsrc_t r, int lane_off, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, byte_off, 0);
}

// filters (COUT, CIN, 3, 3) fp32 -> fragment order [kg][tap][cb][lane][8]: lane (r, h) of block cb holds the weight that takes
// reduction channel ic = 16 kg + 8 h + j to output channel oc = 32 cb + r at tap `tap` of the correlation:
//   forward        W[oc][ic][tap]                  (IC = CIN, OC = COUT)
//   data gradient  W[ic][oc][8 - tap]              (IC = COUT, OC = CIN: transposed, both axes flipped)
__global__ void __launch_bounds__(256) c3_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int IC, int OC, int dgrad) {
    const int i = blockIdx.x * 256 + threadIdx.x, ncb = OC / 32, n = (IC / 16) * 9 * ncb * 64;
    if (i >= n) return;
    const int lane = i & 63, cb = (i >> 6) % ncb, tap = (i / (64 * ncb)) % 9, kg = i / (64 * ncb * 9);
    const int oc = 32 * cb + (lane & 31), ic0 = 16 * kg + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = dgrad ? w[((long)(ic0 + j) * OC + oc) * 9 + 8 - tap] : w[((long)oc * IC + ic0 + j) * 9 + tap];
    wp[i] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

// all eight tables of a step (layers 2-5, forward and data gradient) in one launch: table t = 2 (layer - 2) + dgrad at wp + t * kTabU4
constexpr int kTabU4 = 8 * 9 * 4 * 64;                          // uint4 per table region (the largest: 128 x 64 channels)
struct PackAll { const float* w[4]; };
__global__ void __launch_bounds__(256) c3_pack_all_kernel(const PackAll q, uint4* __restrict__ wp) {
    constexpr int IC[4] = {32, 32, 64, 64}, OC[4] = {32, 64, 64, 128};
    int i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int l = t >> 1, dgrad = t & 1, ic = dgrad ? OC[l] : IC[l], oc = dgrad ? IC[l] : OC[l];
        const int ncb = oc / 32, n = (ic / 16) * 9 * ncb * 64;
        if (i < n) {
            const int lane = i & 63, cb = (i >> 6) % ncb, tap = (i / (64 * ncb)) % 9, kg = i / (64 * ncb * 9);
            const int o = 32 * cb + (lane & 31), ic0 = 16 * kg + 8 * (lane >> 5);
            const float* w = q.w[l];
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dgrad ? w[((long)(ic0 + j) * oc + o) * 9 + 8 - tap] : w[((long)o * ic + ic0 + j) * 9 + tap];
            wp[t * kTabU4 + i] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            return;
        }
        i -= n;
    }
}

// Tile = NI images x TR rows x W columns (NI > 1 only with whole images, TR = H: the 24x24 and 12x12 maps); WGS workgroups per CU.
template <int IC, int OC, int H, int TR, int NI>
struct C3 {
    static constexpr int W = H, NPL = IC / 8, KG = IC / 16, NCB = OC / 32, M = NI * TR * W, NMB = (M + 31) / 32, MBW = (NMB + 3) / 4;
    static constexpr int PITCH = (W + 1) * 16, PR = NI * (TR + 2), PLB = PR * PITCH + 16;
    static constexpr int SLACK = ((128 * MBW - M + W - 1) / W + 3) * PITCH;                  // rows the unused pixel slots' taps may touch
    static constexpr int LDSB = NPL * PLB + SLACK + 256;
    static constexpr int NSLOT = NPL * PR * W, TILES = H / TR;
    static_assert(IC % 16 == 0 && OC % 32 == 0 && H % TR == 0 && (NI == 1 || TR == H) && LDSB <= 160 * 1024, "c3 shape");
};

// y[b][oc][p] = epilogue(sum_{ic,tap} x[b][ic][p + tap - (1,1)] w(oc, ic, tap)); MODE 0: + bias, ReLU; MODE 1: zero where mask <= 0
// (mask may be null); csum (optional): gridDim x OC partial channel sums of y
template <int IC, int OC, int H, int TR, int NI, int MODE, int WGS>
__global__ void __launch_bounds__(256, WGS) c3_kernel(const float* __restrict__ x, const uint4* __restrict__ wp, const float* __restrict__ bias,
                                                      const float* __restrict__ mask, float* __restrict__ y, float* __restrict__ csum, int B) {
    using G = C3<IC, OC, H, TR, NI>;
    constexpr int W = G::W, NCB = G::NCB, MBW = G::MBW, PITCH = G::PITCH, PLB = G::PLB;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p31 = lane & 31;
    for (int i = tid; i < G::LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    int abase[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m) {                             // pixel slot P = (image of the tile, row, column); slots >= M read a valid nowhere
        const int P = 32 * (MBW * wave + m) + p31, im = P / (TR * W), r2 = P - im * (TR * W), yl = r2 / W, xx = r2 - yl * W;
        abase[m] = h * PLB + (im * (TR + 2) + yl) * PITCH + xx * 16;
    }
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, G::KG * 9 * NCB * 1024, 0x00020000);
    // forward: this lane's output channels' biases; data gradient: channel sums of what this lane stores (csum != null)
    float bc[NCB][16];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) bc[cb][r] = MODE == 0 ? bias[32 * cb + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
    const int ntiles = ((B + NI - 1) / NI) * G::TILES;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = (tile / G::TILES) * NI, y0 = (tile % G::TILES) * TR;
        __syncthreads();                                        // the previous tile's reads (and the zero fill) are done
        // gather: slot e = (plane, image, patch row, column); 8 channel planes of the pixel -> one 16-byte slot; four slots
        // (32 loads) per thread in flight
        constexpr int NB = (G::NSLOT + 1023) / 1024;
#pragma unroll 1
        for (int bt = 0; bt < NB; ++bt) {
            float v[4][8];
            int dst[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int e = tid + 256 * (4 * bt + k);
                asm volatile("" : "+v"(e));
                const int pl = e / (G::PR * W), r2 = e - pl * (G::PR * W), ri = r2 / W, xx = r2 - ri * W;
                const int im = ri / (TR + 2), i = ri - im * (TR + 2), yy = y0 - 1 + i, b = b0 + im;
                const bool ok = e < G::NSLOT && (unsigned)yy < (unsigned)H && b < B;
                const float* src = x + (((long)(ok ? b : 0) * IC + 8 * pl) * H + (ok ? yy : 0)) * W + xx;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = src[(long)j * H * W]; v[k][j] = ok ? t : 0.f; }
                dst[k] = e < G::NSLOT ? pl * PLB + ri * PITCH + (xx + 1) * 16 : -1;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (dst[k] >= 0)
                    *(uint4*)(lds + dst[k]) = make_uint4(pack_bf16(v[k][0], v[k][1]), pack_bf16(v[k][2], v[k][3]),
                                                         pack_bf16(v[k][4], v[k][5]), pack_bf16(v[k][6], v[k][7]));
        }
        __syncthreads();

        f32x16_t acc[MBW][NCB];
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][cb][r] = 0.f;
        // rows R = (kg, ky) of three taps; filter fragments one row ahead, pixel fragments one tap ahead
        constexpr int NROW = G::KG * 3;
        u32x4_t wrow[2][3][NCB];
        bf16x8_t a[2][MBW];
        auto toff = [](int t) { const int R = t / 3, kx = t - 3 * R, kg = R / 3, ky = R - 3 * kg; return 2 * kg * PLB + ky * PITCH + kx * 16; };
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) wrow[0][kx][cb] = wload(wr, lane * 16, (kx * NCB + cb) * 1024);
#pragma unroll
        for (int m = 0; m < MBW; ++m) a[0][m] = *(const bf16x8_t*)(lds + abase[m] + toff(0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int R = 0; R < NROW; ++R) {
            const int cur = R & 1;
            if (R + 1 < NROW) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) wrow[cur ^ 1][kx][cb] = wload(wr, lane * 16, (((R + 1) * 3 + kx) * NCB + cb) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int t = R * 3 + kx, ac = t & 1;
                if (t + 1 < NROW * 3) {
#pragma unroll
                    for (int m = 0; m < MBW; ++m) a[ac ^ 1][m] = *(const bf16x8_t*)(lds + abase[m] + toff(t + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MBW; ++m)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[m][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wrow[cur][kx][cb]), a[ac][m],
                                                                             acc[m][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // lanes walk the pixels: 128 contiguous bytes per channel and block
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
            int P = 32 * (MBW * wave + m) + p31;
            asm volatile("" : "+v"(P));                         // (tile-invariant: hipcc would keep all 16 MBW NCB store offsets alive)
            const int im = P / (TR * W), b = b0 + im;
            const bool live = P < G::M && b < B;
            const int o0 = live ? b * (OC * H * W) + y0 * W + (P - im * (TR * W)) : 0;      // (< 2^31 floats: batch <= 1024)
            float gate[NCB][16];                                // the block's mask values first, all in flight (one by one between
            if (MODE == 1 && mask) {                            // the stores they took 2.7x the whole forward kernel)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) gate[cb][r] = mask[o0 + (32 * cb + (r & 3) + 8 * (r >> 2) + 4 * h) * (H * W)];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (live) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int oc = 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float v = acc[m][cb][r];
                        if (MODE == 0) v = fmaxf(v + bc[cb][r], 0.f);
                        else if (mask && !(gate[cb][r] > 0.f)) v = 0.f;
                        y[o0 + oc * (H * W)] = v;
                        if (MODE == 1) bc[cb][r] += v;
                    }
            }
        }
    }
    if (MODE == 1 && csum) {      // csum[workgroup][oc]: butterfly over the 32 pixel lanes, then the four waves in order (the bias gradient of
        __syncthreads();                                        // the layer below when y is the gradient wrt its output)
        float* red = (float*)lds;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = bc[cb][r];
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d);
                if (p31 == 0) red[wave * OC + 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * h] = v;
            }
        __syncthreads();
        if (tid < OC) csum[blockIdx.x * OC + tid] = (red[tid] + red[OC + tid]) + (red[2 * OC + tid] + red[3 * OC + tid]);
    }
}

template <int IC, int OC, int H, int TR, int NI, int MODE, int WGS>
int c3_launch(var_ctx* c, hipStream_t s, const float* x, const float* w, const float* bias, const float* mask, float* y, float* csum,
              int* nparts, int B, void* wp, int prepacked) {
    using G = C3<IC, OC, H, TR, NI>;
    static_assert(WGS * G::LDSB <= 160 * 1024, "c3 LDS per CU");
    const int n = G::KG * 9 * G::NCB * 64;
    static_assert(G::KG * 9 * G::NCB * 64 <= kTabU4, "fragment table region");
    if (!prepacked) {
        hipLaunchKernelGGL(c3_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, (uint4*)wp, IC, OC, MODE);
        VAR_HIP_CHECK(c, hipGetLastError());
    }
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)c3_kernel<IC, OC, H, TR, NI, MODE, WGS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDSB));
        attr |= var_dev_bit(c);
    }
    const int ntiles = ((B + NI - 1) / NI) * G::TILES, cap = 256 * WGS, grid = ntiles < cap ? ntiles : cap;
    if (nparts) *nparts = grid;
    hipLaunchKernelGGL((c3_kernel<IC, OC, H, TR, NI, MODE, WGS>), dim3(grid), dim3(256), G::LDSB, s, x, (const uint4*)wp, bias, mask, y, csum, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// ---- weight gradient -------------------------------------------------------------------------------------------
// dW[co][ci][tap] = sum_{b,p} gy[b][co][p] x[b][ci][p + tap - (1,1)]: k = pixel, 16 consecutive pixels of a row per step; both
// operands come from channel-innermost LDS images (the x patch as above, gy without padding) through the transposing
// ds_read_b64_tr_b16 (plane pitches = 64 mod 256: conflict-free).  Wave w owns the taps w, w+4, w+8 for all (co, ci)
// blocks; the accumulators live across all tiles of the (persistent) workgroup and go to its slab (tap, co, ci) once; a fold
// adds the slabs in workgroup order.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t tr_read2(const unsigned char* p0, const unsigned char* p1) {
    const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p0);
    const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)p1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int CI, int CO, int H, int TR>
struct C3W {
    static constexpr int W = H, NPX = CI / 8, NPG = CO / 8, NCIB = CI / 32, NCOB = CO / 32, KS = TR * W / 16, KPR = W / 16;
    static constexpr int PITCH = (W + 1) * 16;
    static constexpr int XPL0 = (TR + 2) * PITCH + 16, XPL = XPL0 + (64 - XPL0 % 256 + 256) % 256;
    static constexpr int GPL0 = TR * W * 16, GPL = GPL0 + (64 - GPL0 % 256 + 256) % 256;
    static constexpr int XB = NPX * XPL, LDSB = XB + NPG * GPL + 256;
    static constexpr int NXS = NPX * (TR + 2) * W, NGS = NPG * TR * W, TILES = H / TR;
    static_assert(CI % 32 == 0 && CO % 32 == 0 && W % 16 == 0 && H % TR == 0 && XPL % 256 == 64 && GPL % 256 == 64 && 2 * LDSB <= 160 * 1024, "c3w shape");
};

template <int CI, int CO, int H, int TR>
__global__ void __launch_bounds__(256, 2) c3w_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ slab, int B) {
    using G = C3W<CI, CO, H, TR>;
    constexpr int W = G::W, NCIB = G::NCIB, NCOB = G::NCOB, PITCH = G::PITCH;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int g16 = (lane >> 4) & 1, qp = (lane & 15) >> 2, p = lane & 3;
    for (int i = tid; i < G::LDSB / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    // per lane: plane / half-slot of its 4 channels, its pixel 8 h + qp (+ 4) of a k-step
    const int chan = 2 * g16 + (p >> 1), half8 = 8 * (p & 1);
    const int xlane = chan * G::XPL + half8 + (8 * h + qp) * 16;                    // + 4 cib planes, + (row + ky) PITCH + (x0 + kx) 16
    const int glane = G::XB + chan * G::GPL + half8 + (8 * h + qp) * 16;            // + 4 cob planes, + ks 256
    constexpr int NT = 3;                                        // taps wave, wave + 4, wave + 8 (< 9)
    f32x16_t acc[NT][NCOB][NCIB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < NCOB; ++a)
#pragma unroll
            for (int b = 0; b < NCIB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;
    int tapoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { const int tap = min(wave + 4 * t, 8), ky = tap / 3, kx = tap - 3 * ky; tapoff[t] = xlane + ky * PITCH + kx * 16; }
    const int ntaps = wave == 0 ? 3 : 2;

    const int ntiles = B * G::TILES;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / G::TILES, y0 = (tile - b * G::TILES) * TR;
        __syncthreads();
        constexpr int NSL = G::NXS + G::NGS, NB = (NSL + 1023) / 1024;
#pragma unroll 1
        for (int bt = 0; bt < NB; ++bt) {
            float v[4][8];
            int dst[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int e = tid + 256 * (4 * bt + k);
                asm volatile("" : "+v"(e));
                const float* src;
                bool ok;
                if (e < G::NXS) {
                    const int pl = e / ((TR + 2) * W), r2 = e - pl * ((TR + 2) * W), i = r2 / W, xx = r2 - i * W, yy = y0 - 1 + i;
                    ok = (unsigned)yy < (unsigned)H;
                    src = x + (((long)b * CI + 8 * pl) * H + (ok ? yy : 0)) * W + xx;
                    dst[k] = pl * G::XPL + i * PITCH + (xx + 1) * 16;
                } else {
                    const int f = min(e - G::NXS, G::NGS - 1), pl = f / (TR * W), r2 = f - pl * (TR * W);
                    ok = e < NSL;
                    src = gy + (((long)b * CO + 8 * pl) * H + y0) * W + r2;
                    dst[k] = ok ? G::XB + pl * G::GPL + r2 * 16 : -1;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = src[(long)j * H * W]; v[k][j] = ok ? t : 0.f; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (dst[k] >= 0)
                    *(uint4*)(lds + dst[k]) = make_uint4(pack_bf16(v[k][0], v[k][1]), pack_bf16(v[k][2], v[k][3]),
                                                         pack_bf16(v[k][4], v[k][5]), pack_bf16(v[k][6], v[k][7]));
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int row = ks / G::KPR, x0 = 16 * (ks % G::KPR);
            bf16x8_t fa[NCOB], fb[NT][NCIB];
#pragma unroll
            for (int a = 0; a < NCOB; ++a) fa[a] = tr_read2(lds + glane + a * 4 * G::GPL + ks * 256, lds + glane + a * 4 * G::GPL + ks * 256 + 64);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int bb = 0; bb < NCIB; ++bb) {
                    const unsigned char* q = lds + tapoff[t] + bb * 4 * G::XPL + row * PITCH + x0 * 16;
                    fb[t][bb] = tr_read2(q, q + 64);
                }
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t < ntaps) {                                 // uniform per wave
#pragma unroll
                    for (int a = 0; a < NCOB; ++a)
#pragma unroll
                        for (int bb = 0; bb < NCIB; ++bb)
                            acc[t][a][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[t][bb], acc[t][a][bb], 0, 0, 0);
                }
        }
    }
    // slab[wg][tap][co][ci]: lanes walk ci
    float* out = slab + (long)blockIdx.x * 9 * CO * CI + (lane & 31);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = wave + 4 * t;
        if (tap < 9) {
#pragma unroll
            for (int a = 0; a < NCOB; ++a)
#pragma unroll
                for (int bb = 0; bb < NCIB; ++bb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        out[((long)tap * CO + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h) * CI + 32 * bb] = acc[t][a][bb][r];
        }
    }
}

// dW (CO, CI, 3, 3) += the workgroups' slabs (tap, co, ci), in workgroup order: 16 slab elements x 16 slab chains per block
__global__ void __launch_bounds__(256) c3w_fold_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nslabs, int CO, int CI) {
    const int n = 9 * CO * CI, i = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    float a = 0.f;
    if (i < n)
        for (int s = g; s < nslabs; s += 16) a += slab[(long)s * n + i];
    __shared__ float red[16][17];
    red[g][threadIdx.x & 15] = a;
    __syncthreads();
    if (g == 0 && i < n) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
        const int ci = i % CI, co = (i / CI) % CO, tap = i / (CI * CO);
        dw[((long)co * CI + ci) * 9 + tap] += v;
    }
}

template <int CI, int CO, int H, int TR>
int c3w_launch(var_ctx* c, hipStream_t s, const float* x, const float* gy, float* dw, float* slab, int B) {
    using G = C3W<CI, CO, H, TR>;
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)c3w_kernel<CI, CO, H, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDSB));
        attr |= var_dev_bit(c);
    }
    const int ntiles = B * G::TILES, grid = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL((c3w_kernel<CI, CO, H, TR>), dim3(grid), dim3(256), G::LDSB, s, x, gy, slab, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(c3w_fold_kernel, dim3((9 * CO * CI + 15) / 16), dim3(256), 0, s, slab, dw, grid, CO, CI);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// ---- layer 1's weight gradient (3 -> 32 channels at HxH, u8 image) ---------------------------------------------------------
// dW[co][ci][ky][kx] = sum_{b,p} gy[b][co][p] (img[b][ci][p + tap - (1,1)] / 255): k = pixel, 16 consecutive pixels of a row
// per step; rows = co: gy is fp32 NCHW, a lane's 8 pixels are 32 contiguous bytes, loaded straight from HBM a ring of steps
// ahead and rounded; columns = the 27 taps (ci, ky, kx): the image tile sits in LDS as bf16 [ci][row][column + 1] and a tap
// column's 8 consecutive pixels are taken from 5 consecutive dwords with v_perm (the start is odd for kx = 0, 2).  The four
// waves split the tile's k-steps; one slab (32 x 32) per workgroup.
template <int H, int TR>
__global__ void __launch_bounds__(256, 2) c1w_kernel(const unsigned char* __restrict__ img, long bstride, const float* __restrict__ gy,
                                                     float* __restrict__ slab, int B) {
    constexpr int W = H, PITCH = (W + 8) * 2, PLB = (TR + 2) * PITCH, KPR = W / 16, KS = TR * KPR, TILES = H / TR;
    static_assert(W % 16 == 0 && H % TR == 0 && KS % 4 == 0, "c1w shape");
    __shared__ __align__(16) unsigned char lds[3 * PLB + 64];
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
    // this lane's tap column (ci, ky, kx) (columns 27..31 repeat tap 26 and are dropped): element (row + ky, x + kx) of plane ci,
    // elements start one slot in (column -1 = element 1; element 0 is padding for the dword pairing)
    const int col = min(r, 26), ci = col / 9, ky = (col - 9 * ci) / 3, kx = col - 9 * ci - 3 * ky;
    // 8 pixels from x0 + 8 h: elements e0 = x0 + 8 h + kx, e0 + 1, ...: even start (kx = 0, 2) = 4 whole dwords; odd start (kx = 1)
    // = the high half of each dword with the low half of the next
    const int cbase = ci * PLB + ky * PITCH + ((8 * h + kx - (kx == 1 ? 1 : 0)) >> 1) * 4;
    const unsigned sel = kx == 1 ? 0x05040302u : 0x03020100u;
    for (int i = tid; i < (3 * PLB + 64) / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0, 0, 0, 0);
    f32x16_t acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int ntiles = B * TILES;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / TILES, y0 = (tile - b * TILES) * TR;
        __syncthreads();
        for (int e = tid; e < 3 * (TR + 2) * (W / 4); e += 256) {          // 4 pixels per thread: u8 -> /255 -> bf16
            const int pl = e / ((TR + 2) * (W / 4)), r2 = e - pl * ((TR + 2) * (W / 4)), i = r2 / (W / 4), x4 = r2 - i * (W / 4), yy = y0 - 1 + i;
            unsigned v = 0;
            if ((unsigned)yy < (unsigned)H) v = *(const unsigned*)(img + b * bstride + ((long)pl * H + yy) * W + 4 * x4);
            const float f0 = (float)(v & 255u) / 255.f, f1 = (float)((v >> 8) & 255u) / 255.f, f2 = (float)((v >> 16) & 255u) / 255.f,
                        f3 = (float)(v >> 24) / 255.f;
            unsigned short* d = (unsigned short*)(lds + pl * PLB + i * PITCH) + 1 + 4 * x4;       // odd element: 2 + 4 + 2 bytes
            const unsigned p01 = pack_bf16(f0, f1), p23 = pack_bf16(f2, f3);
            d[0] = (unsigned short)p01; *(unsigned*)(d + 1) = (p01 >> 16) | (p23 << 16); d[3] = (unsigned short)(p23 >> 16);
        }
        __syncthreads();
        const float* ga = gy + ((long)b * 32 + r) * H * W + (long)y0 * W + 8 * h;         // + 16 ks
        constexpr int NW = KS / 4, DEPTH = NW < 6 ? NW : 6;                               // this wave's k-steps ks = wave + 4 j
        float4 ring[DEPTH][2];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { const float4* q = (const float4*)(ga + 16 * (wave + 4 * d)); ring[d][0] = q[0]; ring[d][1] = q[1]; }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int ks = wave + 4 * j;
            const float4 a0 = ring[j % DEPTH][0], a1 = ring[j % DEPTH][1];
            if (j + DEPTH < NW) { const float4* q = (const float4*)(ga + 16 * (wave + 4 * (j + DEPTH))); ring[j % DEPTH][0] = q[0]; ring[j % DEPTH][1] = q[1]; }
            const int row = ks / KPR, x0 = 16 * (ks - row * KPR);
            const unsigned* q = (const unsigned*)(lds + cbase + row * PITCH + x0 * 2);
            u32x4_t bq, aq;
            bq.x = __builtin_amdgcn_perm(q[1], q[0], sel); bq.y = __builtin_amdgcn_perm(q[2], q[1], sel);
            bq.z = __builtin_amdgcn_perm(q[3], q[2], sel); bq.w = __builtin_amdgcn_perm(q[4], q[3], sel);
            aq.x = pack_bf16(a0.x, a0.y); aq.y = pack_bf16(a0.z, a0.w); aq.z = pack_bf16(a1.x, a1.y); aq.w = pack_bf16(a1.z, a1.w);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aq), __builtin_bit_cast(bf16x8_t, bq), acc, 0, 0, 0);
        }
    }
    // the four waves' partial sums, in wave order; slab[wg][co][column]
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave][q][lane] = acc[q];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            slab[((long)blockIdx.x * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = (red[0][q][lane] + red[1][q][lane]) + (red[2][q][lane] + red[3][q][lane]);
    }
}

// dW (32, 3, 3, 3) += the slabs (co, column), in workgroup order
__global__ void __launch_bounds__(256) c1w_fold_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nslabs) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;            // i = co * 32 + column
    float a = 0.f;
    for (int s = g; s < nslabs; s += 16) a += slab[(long)s * 1024 + i];
    __shared__ float red[16][17];
    red[g][threadIdx.x & 15] = a;
    __syncthreads();
    if (g == 0) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
        const int co = i >> 5, colm = i & 31;
        if (colm < 27) dw[co * 27 + colm] += v;
    }
}

}  // namespace

long img_bf16_workspace_bytes() { return 8L * kTabU4 * 16 + 256; }       // eight fragment tables (layers 2-5, forward / data gradient)

// the eight tables from the four filters (OIHW fp32) in one launch; afterwards img_bf16_conv(..., prepacked = 1)
int img_bf16_pack_all(var_ctx* c, hipStream_t s, const float* w2, const float* w3, const float* w4, const float* w5, void* ws) {
    PackAll q{{w2, w3, w4, w5}};
    int total = 0;
    constexpr int IC[4] = {32, 32, 64, 64}, OC[4] = {32, 64, 64, 128};
    for (int l = 0; l < 4; ++l) total += ((IC[l] / 16) * 9 * (OC[l] / 32) + (OC[l] / 16) * 9 * (IC[l] / 32)) * 64;      // forward + data gradient
    hipLaunchKernelGGL(c3_pack_all_kernel, dim3((total + 255) / 256), dim3(256), 0, s, q, (uint4*)ws);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// layer = 2 | 3 of the image branch at image side 96 (48 after the first pool); dgrad: x = gy, y = dx, mask = the activation whose
// ReLU gates dx (or null); returns 1 for shapes these kernels do not cover (the caller falls back to the gather-GEMM)
// csum / nparts (optional): *nparts x OC partial channel sums of y -- for a data gradient, the bias gradient of the layer below
int img_bf16_conv(var_ctx* c, hipStream_t s, int layer, int side, int dgrad, const float* x, const float* w, const float* bias,
                  const float* mask, float* y, float* csum, int* nparts, int B, void* ws, int prepacked) {
    void* tab = (uint4*)ws + (2 * (layer - 2) + (dgrad ? 1 : 0)) * kTabU4;      // (layer 2..5 checked below)
#define C3_GO(IC, OC, H, TR, NI, WGS)                                                                                          \
    return dgrad ? c3_launch<OC, IC, H, TR, NI, 1, WGS>(c, s, x, w, bias, mask, y, csum, nparts, B, tab, prepacked)             \
                 : c3_launch<IC, OC, H, TR, NI, 0, WGS>(c, s, x, w, bias, mask, y, csum, nparts, B, tab, prepacked)
    if (layer == 2 && side == 96) { C3_GO(32, 32, 96, 8, 1, 2); }
    if (layer == 3 && side == 48) { C3_GO(32, 64, 48, 8, 1, 2); }
    if (layer == 4 && side == 24) { C3_GO(64, 64, 24, 24, 1, 1); }
    if (layer == 5 && side == 12) { C3_GO(64, 128, 12, 12, 2, 1); }
#undef C3_GO
    return 1;                                                   // not covered
}

// dw (COUT, CIN, 3, 3) += the weight gradient of layer 2 | 3 from its input x and the gradient gy wrt its output (fp32 NCHW);
// slab: 512 x 9*COUT*CIN floats; returns 1 for shapes not covered
int img_bf16_wgrad(var_ctx* c, hipStream_t s, int layer, int side, const float* x, const float* gy, float* dw, float* slab, int B) {
    if (layer == 2 && side == 96) return c3w_launch<32, 32, 96, 4>(c, s, x, gy, dw, slab, B);
    if (layer == 3 && side == 48) return c3w_launch<32, 64, 48, 4>(c, s, x, gy, dw, slab, B);
    return 1;
}

// layer 1's weight gradient from the u8 image batch (bstride bytes between images, 3 planes of side x side first) and the
// gradient gy wrt its output (fp32 NCHW); slab: 512 x 1024 floats; returns 1 for shapes not covered
int img_bf16_wgrad1(var_ctx* c, hipStream_t s, int side, const void* image, long bstride, const float* gy, float* dw, float* slab, int B) {
    if (side != 96) return 1;
    const int ntiles = B * (96 / 8), grid = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL((c1w_kernel<96, 8>), dim3(grid), dim3(256), 0, s, (const unsigned char*)image, bstride, gy, slab, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(c1w_fold_kernel, dim3(1024 / 16), dim3(256), 0, s, slab, dw, grid);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
