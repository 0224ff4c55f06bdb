// Producer/consumer ("wave-specialised") variants of the band-tiled image conv kernels for the big
// layers (the second conv, 43 % of the image MACs): a persistent workgroup of NMW matrix waves and NLW
// loader waves walks its tiles with two LDS buffers.  While the matrix waves run the MFMAs of tile i out
// of buffer i&1, the loader waves pull tile i+1 from HBM/L2 into registers and write it into the other
// buffer; one workgroup barrier per tile hands the buffers over.  HBM latency, the LDS store pass and the
// epilogue of one tile thus all hide behind the matrix work of the other -- which the plain kernels
// (stage, barrier, multiply) cannot do, since all their waves move through the phases together.
#include <stdlib.h>

#include "img_stage.h"

template <int CIN_, int COUT_, int H_, int R_, int NU_, int NMW_, int NLW_>
struct PipeCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = H_, R = R_, NU = NU_, NMW = NMW_, NLW = NLW_;
    static constexpr int HO = (H - 1) / 2 + 1, WO = HO;
    static constexpr int IR = 2 * R + 1;
    static constexpr int PW = 2 * WO + 2;
    static constexpr int PLANE = IR * PW;
    static constexpr int UNIT = CIN * PLANE;
    static constexpr int NB = (HO + R - 1) / R;
    static constexpr int PPU = R * WO;
    static constexpr int NPIX = NU * PPU;
    static constexpr int NPB = (NPIX + 31) / 32;
    static constexpr int NBLK = COUT / 32;
    static constexpr int ITEMS = NPB * NBLK;
    static constexpr int IPW = ITEMS / NMW;
    static constexpr int NW = NMW + NLW;
    static constexpr int BUF = (NU * UNIT + 3) & ~3;
    static constexpr int LDS_FLOATS = 2 * BUF;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
    static_assert(ITEMS % NMW == 0, "matrix waves must share the items evenly");
    static_assert(CIN % 32 == 0 && NU <= 2, "band-tiled layers with 32-multiple input channels, at most 2 units per tile");
};

template <class C>
__global__ void __launch_bounds__(C::NW * 64)
img_conv_fwd_pipe_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                         float* __restrict__ y, int B, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = C::NW * 64, NLT = C::NLW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int total_units = B * C::NB;
    const int ntiles = (total_units + C::NU - 1) / C::NU;
    const int G = gridDim.x;
    const bool loader = wave >= C::NMW;
    constexpr long XB = (long)C::CIN * C::H * C::W;

    // pad columns of both buffers (the loaders write every data cell of a tile)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        lds_zero_cols<NT>(lds + b * C::BUF, C::NU * C::CIN * C::IR, C::PW, 0, 1, tid);
        lds_zero_cols<NT>(lds + b * C::BUF, C::NU * C::CIN * C::IR, C::PW, C::W + 1, C::PW - C::W - 1, tid);
    }

    // ---- loader: register-staged band copies (all loads of the tile in flight, then the LDS stores) ----
    const int ltid = tid - C::NMW * 64;
    using Stager = ColStager<C::CIN, C::H, C::W, C::IR, C::PW, C::PLANE, 1, false, NLT, 1>;
    auto load_tile = [&](int tile, float* buf) {
#pragma unroll
        for (int u = 0; u < C::NU; ++u) {
            const int unit = tile * C::NU + u;
            const bool ok = unit < total_units;
            const int b = ok ? unit / C::NB : 0, band = ok ? unit - b * C::NB : 0;
            Stager::copy(buf + u * C::UNIT, x + (size_t)b * XB, 2 * band * C::R - 1, ok, ltid);
        }
    };

    // ---- matrix-wave state: lane constants of its items (tile-independent) ----
    int pixoff[C::IPW];
    int woff[C::IPW];
    int pix[C::IPW], nbk[C::IPW];
    if (!loader) {
#pragma unroll
        for (int i = 0; i < C::IPW; ++i) {
            const int it = wave + C::NMW * i;
            const int pb = it % C::NPB, nb = it / C::NPB;
            int p = pb * 32 + l31;
            pix[i] = p < C::NPIX ? p : -1;
            if (p >= C::NPIX) p = 0;
            const int u = p / C::PPU, q = p - u * C::PPU;
            const int oyl = q / C::WO, ox = q - oyl * C::WO;
            pixoff[i] = u * C::UNIT + (2 * oyl) * C::PW + 2 * ox + half * C::PLANE;
            woff[i] = nb * 32 + l31 + half * C::COUT;
            nbk[i] = nb;
        }
    }

    // Weights-stationary: the persistent matrix waves keep their whole filter slice (K/2 values per lane
    // and item) in registers for the life of the kernel -- no filter traffic at all inside the tile loop.
    constexpr int SPT = C::CIN / 2;
    constexpr int U = SPT > 16 ? 16 : SPT;
    constexpr int BPT = SPT / U;
    constexpr int NBK = 9 * BPT;
    float wreg[C::IPW][NBK][U];
    if (!loader) {
#pragma unroll
        for (int i = 0; i < C::IPW; ++i)
#pragma unroll
            for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    wreg[i][blk][u] = wp[woff[i] + ((blk / BPT) * C::CIN + 2 * ((blk % BPT) * U + u)) * C::COUT];
    }

    int tile = blockIdx.x;
    if (loader && tile < ntiles) load_tile(tile, lds);
    __syncthreads();
#pragma unroll 1
    for (int it = 0; tile < ntiles; tile += G, ++it) {
        float* cur = lds + (it & 1) * C::BUF;
        if (loader) {
            if (!(dbg & 2) && tile + G < ntiles) load_tile(tile + G, lds + ((it + 1) & 1) * C::BUF);
        } else if (!(dbg & 1)) {
            f32x16 acc[C::IPW];
#pragma unroll
            for (int i = 0; i < C::IPW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            // Input values of block blk+1 are read from LDS into a second register set before the MFMAs of
            // block blk are issued (left to itself hipcc issues each ds_read right before its MFMA and a
            // lone matrix wave then idles an LDS round trip per MFMA).  The filter is already in registers.
            float bbuf[2][C::IPW][U];
            auto fetch = [&](int buf, int blk) {
                const int tap = blk / BPT, c2b = (blk % BPT) * U;
                const int toff = (tap / 3) * C::PW + (tap % 3);
#pragma unroll
                for (int i = 0; i < C::IPW; ++i)
#pragma unroll
                    for (int u = 0; u < U; ++u) bbuf[buf][i][u] = cur[pixoff[i] + 2 * (c2b + u) * C::PLANE + toff];
            };
            fetch(0, 0);
#pragma unroll
            for (int blk = 0; blk < NBK; ++blk) {
                if (blk + 1 < NBK) fetch((blk + 1) & 1, blk + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int i = 0; i < C::IPW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[i][blk][u], bbuf[blk & 1][i][u], acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: bias + ReLU, NCHW store
#pragma unroll
            for (int i = 0; i < C::IPW; ++i) {
                const int p = pix[i];
                if (p < 0) continue;
                const int u = p / C::PPU, q = p - u * C::PPU;
                const int unit = tile * C::NU + u;
                if (unit >= total_units) continue;
                const int b = unit / C::NB, band = unit - b * C::NB;
                const int oy = band * C::R + q / C::WO;
                if (oy >= C::HO) continue;
                float* yp = y + (size_t)b * C::COUT * C::HO * C::WO + band * C::R * C::WO + q;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nbk[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = acc[i][r] + bias[n];
                    yp[(size_t)n * C::HO * C::WO] = v > 0.f ? v : 0.f;
                }
            }
        }
        __syncthreads();                               // buffer hand-over
    }
}

//                   CIN COUT  H  R NU NMW NLW
using P84_2 = PipeCfg<32, 32, 42, 3, 2, 4, 4>;
using P96_2 = PipeCfg<32, 32, 48, 4, 1, 3, 3>;

template <class C>
static int launch_pipe(var_ctx* c, hipStream_t s, const float* x, const float* wp, const float* bias, float* y, int B) {
    static bool attr_set = false;
    if (!attr_set) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)img_conv_fwd_pipe_kernel<C>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const int ntiles = (B * C::NB + C::NU - 1) / C::NU;
    const int per_cu = C::LDS_BYTES <= 80 * 1024 ? 2 : 1;
    const int G = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    hipLaunchKernelGGL(img_conv_fwd_pipe_kernel<C>, dim3(G), dim3(C::NW * 64), C::LDS_BYTES, s, x, wp, bias, y, B, getenv("VAR_DBG") ? atoi(getenv("VAR_DBG")) : 0);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// second image conv (32 -> 32 channels) forward, pipelined
int launch_img_fwd_conv2_pipe(var_ctx* c, hipStream_t s, const float* x, const float* wp, const float* bias,
                              float* y, int B) {
    ProfScope prof(c, s, TAG_IMG_FWD0 + 1);
    return c->H == 84 ? launch_pipe<P84_2>(c, s, x, wp, bias, y, B) : launch_pipe<P96_2>(c, s, x, wp, bias, y, B);
}
