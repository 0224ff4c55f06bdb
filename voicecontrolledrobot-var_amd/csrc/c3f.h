// fp32 3x3 / stride 1 / pad 1 convolution + bias + ReLU (+ 2x2 max pool) for SMALL batches: the five big layers of the
// actor-critic's image stack (models/RL/arm_RL_model.py:20-31, conv 2..6 of the 96x96 branch) at the RL stage's 8 envs.
// The gather-GEMM of gg.h computes im2col addresses per element and, at 8 images, needs split-K slabs, a finish launch
// and a pool launch per layer (199 us for these five layers).  Here a workgroup owns a BAND of TR rows of one image and
// 16 * NCBW output channels:
//   * the input band (TR + 2 rows, every input channel, zero halo) is staged ONCE into LDS with 16-byte loads; a tap is
//     an immediate offset of a `ds_read_b32`, the plane pitch is 16 mod 32 floats so that the 16 pixels x 4 channels of
//     a B operand hit 64 distinct banks;
//   * v_mfma_f32_16x16x4_f32, A = filter (16 output channels x 4 input channels), B = 16 consecutive pixels of the band;
//     a wave keeps NPB pixel blocks x NCB channel blocks of accumulators (4 VGPRs each), so one A operand feeds NPB and one
//     B read feeds NCB matrix instructions;
//   * the filter is packed in MFMA A-fragment order (pack_item(): 16 bytes = the A operands of four k-steps) and travels
//     through LDS in 16-channel chunks beside the band's (a wave streaming it from L2 one tap ahead waited ~20 cycles per
//     matrix instruction: 55 instead of 32 cycles each);
//   * the epilogue adds the bias, applies ReLU and either stores NCHW or goes through an LDS tile for the 2x2 max pool (the
//     un-pooled map is never written: the acting path has no backward).
#pragma once
#include "var_common.h"
#ifndef VAR_PH_C3F_CIN
#define VAR_PH_C3F_CIN 128      // which layer `make phases` times (tools/phases_arm.py, phases 16..21)
#define VAR_PH_C3F_COUT 128
#endif

namespace c3f {
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CIN_, int COUT_, int H_, int TR_, int NCBW_, int WCB_, bool POOL_>
struct Cfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, TR = TR_, NCBW = NCBW_, WCB = WCB_;
    static constexpr bool POOL = POOL_;
    static constexpr int PW = W + 8, ROWS = TR + 2;                     // pixel x of a row sits at float 4 + x
    static constexpr int PLANE0 = ROWS * PW;
    static constexpr int PLANE = PLANE0 + (48 - PLANE0 % 32) % 32;      // == 16 (mod 32)
    static constexpr int NPX = TR * W, PBT = NPX / 16;                  // pixel blocks of the band
    static constexpr int WPB = 4 / WCB;                                 // waves along the pixel blocks
    static constexpr int NPB = (PBT + WPB - 1) / WPB, NCB = NCBW / WCB;
    static constexpr int KG = CIN / 16;
    static constexpr int BAND = CIN * PLANE;
    static constexpr int OP = NPX + 4;                                  // pool tile pitch: 4 (mod 8)
    static constexpr int OUTT = POOL ? 16 * NCBW * OP : 0;
    static constexpr int ACH = NCBW * 9 * 64 * 4;                       // floats of one 16-channel filter chunk (A fragments)
    static constexpr int BOT = BAND > OUTT ? BAND : OUTT;               // band, later the pool tile
    static constexpr int LDSF = BOT + 2 * ACH;                          // + two filter chunks (double buffer)
    static constexpr int BANDS = H / TR, CBG = COUT / (16 * NCBW);
    static constexpr int WFLOATS = COUT * CIN * 9;
    static_assert(PLANE % 32 == 16 && PLANE % 4 == 0, "plane pitch");
    static_assert(CIN % 16 == 0 && COUT % (16 * NCBW) == 0 && H % TR == 0 && NPX % 16 == 0 && W % 4 == 0, "c3f shape");
    static_assert(NCBW % WCB == 0 && 4 % WCB == 0 && (!POOL || TR % 2 == 0), "c3f wave split");
    static_assert(LDSF * 4 <= 160 * 1024, "c3f LDS");
    static_assert((12 * PLANE + 2 * PW + 5) * 4 < 65536, "ds_read immediate offset");
};

// wp[((cb * KG + kg) * 9 + tap) * 64 + lane] = { w[16 cb + (lane & 15)][16 kg + 4 j + (lane >> 4)][tap] : j = 0..3 }, all layers of
// a stack in one launch (the parameters may change between two forwards -- PPO updates them in PyTorch --, so the acting path
// re-packs per call: 1.1 MB, one launch)
constexpr int kMaxLayers = 8;
struct PackDesc {
    int n_layers;
    int w_off[kMaxLayers], wp_off[kMaxLayers] /* float4 units */, cin[kMaxLayers], cout[kMaxLayers], first[kMaxLayers + 1];
};
__device__ __forceinline__ void pack_item(const float* __restrict__ params, f32x4* __restrict__ wp, const PackDesc& d, int i) {
    if (i >= d.first[d.n_layers]) return;
    int l = 0;
    while (i >= d.first[l + 1]) ++l;
    i -= d.first[l];
    const int CIN = d.cin[l], KG = CIN / 16;
    const int lane = i & 63, tap = (i >> 6) % 9, kg = (i / (64 * 9)) % KG, cb = i / (64 * 9 * KG);
    const float* q = params + d.w_off[l] + ((long)(16 * cb + (lane & 15)) * CIN + 16 * kg + (lane >> 4)) * 9 + tap;
    wp[d.wp_off[l] + i] = f32x4{q[0], q[4 * 9], q[8 * 9], q[12 * 9]};
}

// ---- conv 1 of the stack (3 -> 32, 3x3 pad 1, 96x96; u8 / 255 or float input) + the filter pack of the later layers in ONE launch:
// workgroups [0, nconv) own bands of two image rows (K = 27 padded to 28 = 7 steps of v_mfma_f32_16x16x4_f32, the filter read in
// place from its state_dict() layout, a per-lane table turns k = 4 j + (lane >> 4) into its (channel, tap) offset in the LDS
// band), the rest run pack_item().  The layer is a 9.4-MB write: the point is one launch instead of two on a dependent chain.
constexpr int C1_H = 96, C1_TR = 2, C1_PW = C1_H + 8, C1_ROWS = C1_TR + 2, C1_PLANE = C1_ROWS * C1_PW, C1_BANDS = C1_H / C1_TR;
template <bool U8>
__global__ void __launch_bounds__(256) c1f_pack_kernel(const void* __restrict__ image, long bstride, const float* __restrict__ params, int w_off,
                                                      int b_off, float* __restrict__ y, int nconv, f32x4* __restrict__ wp, PackDesc d) {
    if ((int)blockIdx.x >= nconv) { pack_item(params, wp, d, ((int)blockIdx.x - nconv) * 256 + threadIdx.x); return; }
    constexpr int H = C1_H, W = C1_H, PW = C1_PW, ROWS = C1_ROWS, PLANE = C1_PLANE, NPB = (C1_TR * W / 16) / 4;
    __shared__ float band[3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.x / C1_BANDS, r0 = (blockIdx.x - b * C1_BANDS) * C1_TR;
    const float* w = params + w_off;
    // A operands: a[cb][j] = w[16 cb + l15][k = 4 j + lk], zero for the padding k = 27; B offsets of the same k
    float a[2][7];
    int off[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = 4 * j + lk, kc = k < 27 ? k : 26;
        const int ch = kc / 9, tap = kc - 9 * ch, dy = tap / 3, dx = tap - 3 * dy;
        off[j] = ch * PLANE + dy * PW + dx + 3;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) { const float v = w[(16 * cb + l15) * 27 + kc]; a[cb][j] = k < 27 ? v : 0.f; }
    }
    // band: rows r0 - 1 .. r0 + TR of the three channels, zeros outside the image
    {
        constexpr int N = 3 * ROWS * W, PER = (N + 255) / 256;
        float v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            int e = tid + 256 * u;
            e = e < N ? e : N - 1;
            const int ch = e / (ROWS * W), rem = e - ch * (ROWS * W), br = rem / W, col = rem - br * W;
            const int ir = r0 - 1 + br, irc = ir < 0 ? 0 : (ir >= H ? H - 1 : ir);
            const long o = (long)b * bstride + ((long)ch * H + irc) * W + col;
            v[u] = U8 ? (float)((const uint8_t*)image)[o] / 255.f : ((const float*)image)[o];          // dataset.py:67-68
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            if (e < N) {
                const int ch = e / (ROWS * W), rem = e - ch * (ROWS * W), br = rem / W, col = rem - br * W;
                const int ir = r0 - 1 + br;
                band[ch * PLANE + br * PW + 4 + col] = (ir < 0 || ir >= H) ? 0.f : v[u];
            }
        }
        if (tid < 3 * ROWS) { band[(tid / ROWS) * PLANE + (tid % ROWS) * PW + 3] = 0.f; band[(tid / ROWS) * PLANE + (tid % ROWS) * PW + 4 + W] = 0.f; }
    }
    __syncthreads();
    const float* bias = params + b_off;
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int p = (wave * NPB + i) * 16 + l15, row = p / W, col = p - row * W;
        const float* xp = band + row * PW + col;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float bv = xp[off[j]];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][j], bv, acc[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * cb + 4 * lk + r;
                const float v = acc[cb][r] + bias[co];
                y[(((size_t)b * 32 + co) * H + r0 + row) * W + col] = v > 0.f ? v : 0.f;
            }
    }
}

template <class C>
__global__ void __launch_bounds__(256) c3f_kernel(const float* __restrict__ x, const f32x4* __restrict__ wp, const float* __restrict__ bias,
                                                 float* __restrict__ y, int B) {
    constexpr int CIN = C::CIN, COUT = C::COUT, H = C::H, W = C::W, TR = C::TR, PW = C::PW, ROWS = C::ROWS, PLANE = C::PLANE;
    constexpr int NPB = C::NPB, NCB = C::NCB, KG = C::KG, PBT = C::PBT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int tile = blockIdx.x % (B * C::BANDS), cbg = blockIdx.x / (B * C::BANDS);
    const int b = tile / C::BANDS, r0 = (tile - b * C::BANDS) * TR;
    const int wc = wave % C::WCB, wpb = wave / C::WCB;
    const int cb0 = cbg * C::NCBW + wc * NCB;                      // first 16-channel block of this wave
    PHR_INIT(0, 0);

    // ---- stage the band: rows r0 - 1 .. r0 + TR, zeros outside the image, in chunks of 16 input channels (= one outer step
    //      of the product loop): chunk 0 before the loop, chunk kg + 1 is in flight (registers) during the products of chunk kg
    //      The filter chunk of the same 16 channels (9 taps x NCBW channel blocks of A fragments, shared by the four waves)
    //      travels with it into one of two LDS buffers.
    constexpr int Q = W / 4, CSLOTS = 16 * ROWS * Q, NLD = (CSLOTS + 255) / 256;
    constexpr int ASLOTS = C::NCBW * 9 * 64, NLA = (ASLOTS + 255) / 256;
    const float* xb = x + (size_t)b * CIN * H * W;
    float* aw = lds + C::BOT;
    float4 sv[NLD];
    f32x4 sa[NLA];
    auto issue = [&](int kg) {
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            int e = tid + 256 * u;
            e = e < ASLOTS ? e : ASLOTS - 1;
            const int cbl = e / 576, rem = e - cbl * 576;
            sa[u] = wp[((size_t)(cbg * C::NCBW + cbl) * KG + kg) * 576 + rem];
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            int e = tid + 256 * u;
            e = e < CSLOTS ? e : CSLOTS - 1;
            const int cl = e / (ROWS * Q), rem = e - cl * (ROWS * Q), br = rem / Q, q = rem - br * Q;
            const int ir = r0 - 1 + br, irc = ir < 0 ? 0 : (ir >= H ? H - 1 : ir);
            sv[u] = *(const float4*)(xb + ((size_t)(16 * kg + cl) * H + irc) * W + 4 * q);     // rows outside the image: zeroed in commit()
        }
    };
    auto commit = [&](int kg) {
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            const int e = tid + 256 * u;
            if (e < ASLOTS) *(f32x4*)(aw + (kg & 1) * C::ACH + 4 * e) = sa[u];
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = tid + 256 * u;
            if (e < CSLOTS) {
                const int cl = e / (ROWS * Q), rem = e - cl * (ROWS * Q), br = rem / Q, q = rem - br * Q;
                const int ir = r0 - 1 + br;
                // (selecting the zeros where the load is issued would make issue() wait for its own loads)
                *(float4*)(lds + (16 * kg + cl) * PLANE + br * PW + 4 + 4 * q) = (ir < 0 || ir >= H) ? float4{0.f, 0.f, 0.f, 0.f} : sv[u];
            }
        }
    };
    issue(0);
    for (int e = tid; e < CIN * ROWS; e += 256) {                  // halo columns x = -1 and x = W of every row
        const int ch = e / ROWS, br = e - ch * ROWS;
        lds[ch * PLANE + br * PW + 3] = 0.f;
        lds[ch * PLANE + br * PW + 4 + W] = 0.f;
    }
    commit(0);
    PHR(0);
    if (KG > 1) issue(1);
    __builtin_amdgcn_sched_barrier(0);
    // lane constants: where this lane's pixel of each block sits in the band (row-major over TR x W), plus its k plane
    int va[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        int pb = wpb * NPB + i;
        pb = pb < PBT ? pb : PBT - 1;
        const int p = pb * 16 + l15, row = p / W, col = p - row * W;
        va[i] = lk * PLANE + row * PW + col + 3;
    }
    f32x4 acc[NPB][NCB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    PHR(1);

    // ---- products: K = (16 input channels) x (9 taps) per outer step
#pragma unroll 1
    for (int kg = 0; kg < KG; ++kg) {
        const float* xk = lds + kg * 16 * PLANE;
        const f32x4* ak = (const f32x4*)(aw + (kg & 1) * C::ACH) + (wc * NCB) * 576 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            f32x4 a[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) a[cb] = ak[cb * 576 + tap * 64];
            if (tap == 3 && kg + 1 < KG) {
                // the next chunk (in registers since the middle of the previous block) goes to LDS and the one after it leaves
                // for the registers HERE, between matrix instructions that are already queued, not at the block's barrier
                commit(kg + 1);
                if (kg + 2 < KG) issue(kg + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bv[NPB];
#pragma unroll
                for (int i = 0; i < NPB; ++i) bv[i] = xk[va[i] + 4 * j * PLANE + dy * PW + dx];
#pragma unroll
                for (int i = 0; i < NPB; ++i)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][j], bv[i], acc[i][cb], 0, 0, 0);
            }
        }
        PHR(2);
        // LDS-only barrier: __syncthreads() also drains vmcnt, i.e. waits for the chunk that has just been requested
        if (kg + 1 < KG) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        PHR(3);
    }

    // ---- epilogue: D[channel 4 (lane >> 4) + r][pixel lane & 15]
    if (C::POOL) __syncthreads();                                   // every wave is done with the band: the pool tile reuses it
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int pb = wpb * NPB + i;
        if (pb >= PBT) continue;                                    // wave-uniform
        const int p = pb * 16 + l15, row = p / W, col = p - row * W;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * (cb0 + cb) + 4 * lk + r;
                float v = acc[i][cb][r] + bias[co];
                v = v > 0.f ? v : 0.f;
                if (C::POOL) lds[(16 * (wc * NCB + cb) + 4 * lk + r) * C::OP + p] = v;
                else y[(((size_t)b * COUT + co) * H + r0 + row) * W + col] = v;
            }
    }
    PHR(4);
    if (C::POOL) {
        __syncthreads();
        constexpr int HP = H / 2, WP = W / 2, TP = TR / 2, NOUT = 16 * C::NCBW * TP * WP;
        for (int e = tid; e < NOUT; e += 256) {
            const int cl = e / (TP * WP), rem = e - cl * (TP * WP), pr = rem / WP, pc = rem - pr * WP;
            const float* q = lds + cl * C::OP + (2 * pr) * W + 2 * pc;
            const float m = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[W], q[W + 1]));
            y[(((size_t)b * COUT + 16 * cbg * C::NCBW + cl) * HP + r0 / 2 + pr) * WP + pc] = m;
        }
    }
    PHR(5);
#ifdef VAR_PHASES
    if (ph_on && CIN == VAR_PH_C3F_CIN && COUT == VAR_PH_C3F_COUT)
        for (int i_ = 0; i_ < 8; ++i_) g_phase[16 + i_] += ph_a[i_];
#endif
}

template <class C>
int launch(var_ctx* c, hipStream_t s, const float* x, const f32x4* wp, const float* bias, float* y, int B) {
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)c3f_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDSF * 4));
        attr |= var_dev_bit(c);
    }
    hipLaunchKernelGGL(c3f_kernel<C>, dim3(B * C::BANDS * C::CBG), dim3(256), C::LDSF * 4, s, x, wp, bias, y, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// ---- small convolutions of a small batch: the actor-critic's two last layers (conv 7: 128 -> 256, 3x3 stride 2 pad 0, 12x12 ->
// 5x5; conv 8: 256 -> 128, 3x3, 5x5 -> 3x3: a quarter of a GFLOP, 2.4 MB of filter) and conv 3..5 of the Kuka encoder when it runs
// frozen on the RL stage's 8 images (3x3 stride 2 pad 1, 21 -> 11 -> 6 -> 3).  A workgroup = (image, 16 output channels): the
// image's input (<= 74 KB, zero halo for PAD = 1) is staged in LDS, the four waves split K in 16-channel groups (WK ways) and the
// pixel blocks (4 / WK ways), every wave requests ALL of its filter share before the staging (<= 36 loads in flight per lane;
// RAW: read in place from the state_dict() layout, four 4-byte loads instead of one packed 16-byte load), the WK partial tiles
// are folded through LDS in wave order.  Replaces a split-K gather-GEMM launch + finish per layer (actor-critic) / the
// one-workgroup-per-image conv 3-5 kernel whose time is the per-image latency (frozen encoder: 34 us at 8 images).
template <int CIN_, int COUT_, int HIN_, int S_, int HOUT_, int PAD_ = 0, bool RAW_ = false>
struct SmallCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, S = S_, HOUT = HOUT_, PAD = PAD_;
    static constexpr bool RAW = RAW_;
    static constexpr int NPX = HOUT * HOUT, NPBT = (NPX + 15) / 16, KG = CIN / 16;
    static constexpr int WK = KG >= 4 ? 4 : KG, WP = 4 / WK;              // waves along K / along the pixel blocks
    static constexpr int KGW = KG / WK, NPB = (NPBT + WP - 1) / WP;
    static constexpr int HP = HIN + 2 * PAD;                               // padded input side
    static constexpr int PL0 = HP * HP, PL = PL0 + (48 - PL0 % 32) % 32;   // == 16 (mod 32)
    static constexpr int IMG = CIN * PL, RED = 4 * NPB * 4 * 64;
    static constexpr int LDSF = IMG + RED;
    static_assert(KG % WK == 0 && 4 % WK == 0 && COUT % 16 == 0 && LDSF * 4 <= 160 * 1024, "small conv shape");
    static_assert((HOUT - 1) * S + 3 <= HP, "taps stay inside the (padded) input");
};

template <class C>
__global__ void __launch_bounds__(256) c3s_kernel(const float* __restrict__ x, const void* __restrict__ wsrc, const float* __restrict__ bias,
                                                 float* __restrict__ y, int B) {
    constexpr int CIN = C::CIN, COUT = C::COUT, HIN = C::HIN, S = C::S, HOUT = C::HOUT, NPX = C::NPX, NPB = C::NPB, KG = C::KG, PL = C::PL;
    constexpr int HP = C::HP, PAD = C::PAD, WK = C::WK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* red = lds + C::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.x % B, cb = blockIdx.x / B;
    const int wk = wave % WK, wp = wave / WK;
    // this wave's share of the filter: groups kg = wk, wk + WK, ...
    f32x4 a[C::KGW][9];
    if (C::RAW) {
        const float* w = (const float*)wsrc + ((long)(16 * cb + l15) * CIN + lk) * 9;
#pragma unroll
        for (int g = 0; g < C::KGW; ++g)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* q = w + (16 * (wk + WK * g)) * 9 + tap;
                a[g][tap] = f32x4{q[0], q[4 * 9], q[8 * 9], q[12 * 9]};
            }
    } else {
        const f32x4* wa = (const f32x4*)wsrc + (size_t)cb * KG * 576 + lane;
#pragma unroll
        for (int g = 0; g < C::KGW; ++g)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a[g][tap] = wa[((wk + WK * g) * 9 + tap) * 64];
    }
    // the image -> LDS [channel][PL] (padded side HP with a zero ring)
    if (PAD) {      // only the one-cell ring around each plane: the interior is written below (disjoint cells, one barrier for both)
        constexpr int RING = 4 * HP - 4;
        for (int e = tid; e < CIN * RING; e += 256) {
            const int ch = e / RING, q = e - ch * RING;
            const int yy = q < HP ? 0 : (q < 2 * HP ? HP - 1 : 1 + (q - 2 * HP) / 2);
            const int xx = q < HP ? q : (q < 2 * HP ? q - HP : ((q - 2 * HP) & 1 ? HP - 1 : 0));
            lds[ch * PL + yy * HP + xx] = 0.f;
        }
    }
    {
        constexpr int P0 = HIN * HIN, N4 = CIN * P0 / 4;
        static_assert((CIN * P0) % 4 == 0, "16-byte loads over an image");
        const float4* src = (const float4*)(x + (size_t)b * CIN * P0);
#pragma unroll 1
        for (int e0 = tid; e0 < N4; e0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; v[u] = src[e < N4 ? e : N4 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u;
                if (e < N4) {
                    const float t[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = 4 * e + q, ch = idx / P0, pos = idx - ch * P0, yy = pos / HIN, xx = pos - yy * HIN;
                        lds[ch * PL + (yy + PAD) * HP + xx + PAD] = t[q];
                    }
                }
            }
        }
    }
    int va[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        int p = (wp * NPB + i) * 16 + l15;
        p = p < NPX ? p : NPX - 1;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        va[i] = lk * PL + oy * S * HP + ox * S;
    }
    f32x4 acc[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int g = 0; g < C::KGW; ++g) {
        const float* xk = lds + (wk + WK * g) * 16 * PL;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NPB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][tap][j], xk[va[i] + 4 * j * PL + dy * HP + dx], acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NPB + i) * 4 + r) * 64 + lane] = acc[i][r];
    __syncthreads();
    // fold the WK partial tiles of a pixel block (waves wp * WK .. + WK - 1) in wave order
    for (int e = tid; e < C::WP * NPB * 4 * 64; e += 256) {
        const int l = e & 63, r = (e >> 6) & 3, i = (e >> 8) % NPB, wq = e / (256 * NPB);
        float sum = red[(((wq * WK) * NPB + i) * 4 + r) * 64 + l];
#pragma unroll
        for (int k = 1; k < WK; ++k) sum += red[(((wq * WK + k) * NPB + i) * 4 + r) * 64 + l];
        const int co = 16 * cb + 4 * (l >> 4) + r, px = (wq * NPB + i) * 16 + (l & 15);
        if (px < NPX) {
            const float v = sum + bias[co];
            y[((size_t)b * COUT + co) * NPX + px] = v > 0.f ? v : 0.f;
        }
    }
}

template <class C>
int launch_small(var_ctx* c, hipStream_t s, const float* x, const void* wsrc, const float* bias, float* y, int B) {
    static unsigned attr = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)c3s_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDSF * 4));
        attr |= var_dev_bit(c);
    }
    hipLaunchKernelGGL(c3s_kernel<C>, dim3(B * (C::COUT / 16)), dim3(256), C::LDSF * 4, s, x, wsrc, bias, y, B);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
}  // namespace c3f
