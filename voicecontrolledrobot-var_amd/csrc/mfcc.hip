// Audio front-end on the GPU: int16 PCM -> (frames, 40) MFCC, the work the reference does per
// item inside its DataLoader workers with torchaudio.transforms.MFCC
// (Envs/audioLoader.py:147-157) followed by processSoundFeat (Envs/audioLoader.py:241-252):
//   x/32768 -> reflect-padded frames (n_fft 512, hop 160) x periodic Hamming(400) centred in 512
//   -> |rFFT|^2 -> 40 HTK-mel triangles -> log(. + 1e-6) -> orthonormal DCT-II -> (T, 40),
//   T = 1 + N/160; rows beyond T are zero, rows beyond out_frames are dropped.
// One wavefront per frame: the 512-point real FFT is a 256-point complex Stockham radix-4 FFT
// (4 stages, exactly one radix-4 butterfly per lane per stage, ping-pong in LDS) plus the
// even/odd split; a workgroup of 4 waves walks the frames of one clip.  PCM is read as
// coalesced 128-byte rows straight from HBM/L2 (each sample is touched by 2.5 frames).
// Tables (window, twiddles, mel triangles, DCT) are computed once on the host in double.
#include <math.h>

#include <vector>

#include "var_common.h"

namespace {
constexpr int NFFT = 512, WIN = 400, HOP = 160, NMEL = 40, NMFCC = 40, NFREQ = 257;
constexpr int WOFF = (NFFT - WIN) / 2;   // 56

// table layout (floats)
constexpr int TB_WIN = 0;                 // 400
constexpr int TB_TW256 = 400;             // 256 x (cos, sin) of -2 pi m / 256
constexpr int TB_TW512 = TB_TW256 + 512;  // 256 x (cos, sin) of -2 pi k / 512
constexpr int TB_DCT = TB_TW512 + 512;    // [n][k] 40 x 40
constexpr int TB_MSTART = TB_DCT + 1600;  // 40 ints
constexpr int TB_MCOUNT = TB_MSTART + 40;
constexpr int TB_MOFF = TB_MCOUNT + 40;
constexpr int TB_MW = TB_MOFF + 40;       // <= 640 weights (triangles back to back; host-side only)
constexpr int MAXC = 34;                  // trip count of the mel loop (widest triangle: 33 bins)
constexpr int TB_MWD = TB_MW + 640;       // dense [q][40]: weight of bin mstart[m] + q in filter m, 0 beyond its triangle
constexpr int TB_WIN512 = TB_MWD + MAXC * 40;   // 512: Hamming(400) / 32768 centred in the frame, 0 outside
constexpr int TB_TOTAL = TB_WIN512 + 512; // a multiple of 4

struct cplx { float x, y; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }

// Each wave owns its frames end to end and never synchronises with the other waves: all exchange
// goes through the wave's private LDS region, where a wave's own DS operations execute in order,
// so a compiler-level wave barrier is all that separates a butterfly stage from the next.
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()

constexpr int MW = 16;     // waves per workgroup (one clip per workgroup, frames dealt to waves)

__global__ void __launch_bounds__(MW * 64)
mfcc_kernel(const int16_t* __restrict__ pcm, const int* __restrict__ lens, const int* __restrict__ clip_index,
            int pcm_stride, int out_frames, const float* __restrict__ tab, float* __restrict__ out) {
    __shared__ float tabs[TB_TOTAL];                 // window, twiddles, DCT, mel triangles (15 KB)
    __shared__ cplx bufA[MW][256];
    __shared__ cplx bufB[MW][256];                   // after the FFT: power spectrum (264 floats) + log-mel (40)
    const int clip = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < TB_TOTAL / 4; e += MW * 64) ((float4*)tabs)[e] = ((const float4*)tab)[e];
    const int N = lens[clip];
    const int T = 1 + N / HOP;
    const int16_t* sig = pcm + (size_t)(clip_index ? clip_index[clip] : clip) * pcm_stride;
    const int* itab = (const int*)tabs;
    const cplx* tw256 = (const cplx*)(tabs + TB_TW256);
    const cplx* tw512 = (const cplx*)(tabs + TB_TW512);
    float* xs = (float*)bufA[wave];
    __syncthreads();
    // mel filter of this lane (lanes >= 40 idle in that phase)
    const int ml = lane < NMEL ? lane : 0;
    const int mst = itab[TB_MSTART + ml];
    float dct[NMEL];
#pragma unroll
    for (int n = 0; n < NMEL; ++n) dct[n] = tabs[TB_DCT + n * NMFCC + ml];

    float* pw = (float*)bufB[wave];                  // the FFT result ends in bufA (4 stages), bufB is free then
    float* lm = pw + 264;
    for (int t = wave; t < out_frames; t += MW) {
        // frames beyond T are MFCC-domain zero padding; N == 0 is the "empty" class (dataset.py:37-38)
        const bool live = t < T && N > 0;
        if (!live) {
            if (lane < NMFCC) out[((size_t)clip * out_frames + t) * NMFCC + lane] = 0.f;
            continue;
        }
        // 1. windowed frame -> LDS.  Frames whose 512 samples all lie inside the clip (all but two at either end)
        //    take two samples per 4-byte load and the zero-extended window table: no index logic per sample.
        const int p0 = t * HOP - NFFT / 2;
        if (p0 >= 0 && p0 + NFFT <= N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = lane + 64 * i;
                const uint32_t two = *(const uint32_t*)(sig + p0 + 2 * n);        // p0 even, clip rows 4-byte aligned
                const float2 w = *(const float2*)(tabs + TB_WIN512 + 2 * n);
                float2 z;
                z.x = (float)(int16_t)(two & 0xffff) * w.x;
                z.y = (float)((int32_t)two >> 16) * w.y;
                *(float2*)(xs + 2 * n) = z;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = lane + 64 * i;
                float v = 0.f;
                if (idx >= WOFF && idx < WOFF + WIN) {
                    int pos = p0 + idx;
                    if (pos < 0) pos = -pos;
                    if (pos >= N) pos = 2 * (N - 1) - pos;
                    pos = pos < 0 ? 0 : (pos >= N ? N - 1 : pos);
                    v = (float)sig[pos] * tabs[TB_WIN512 + idx];
                }
                xs[idx] = v;
            }
        }
        WAVE_SYNC();
        // 2. 256-point complex FFT of z[n] = x[2n] + i x[2n+1]: Stockham radix-4, natural order out
        cplx* src = bufA[wave];
        cplx* dst = bufB[wave];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int Ns = 1 << (2 * s);
            const int jm = lane & (Ns - 1);
            const int m = jm * (64 / Ns);
            cplx v0 = src[lane], v1 = src[lane + 64], v2 = src[lane + 128], v3 = src[lane + 192];
            if (s > 0) {
                v1 = cmul(v1, tw256[m]);
                v2 = cmul(v2, tw256[(2 * m) & 255]);
                v3 = cmul(v3, tw256[(3 * m) & 255]);
            }
            const cplx a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3);
            const cplx d = csub(v1, v3);
            const cplx a3 = {d.y, -d.x};                   // (v1 - v3) * (-i)
            const int idx = (lane / Ns) * Ns * 4 + jm;
            dst[idx] = cadd(a0, a2);
            dst[idx + Ns] = cadd(a1, a3);
            dst[idx + 2 * Ns] = csub(a0, a2);
            dst[idx + 3 * Ns] = csub(a1, a3);
            WAVE_SYNC();
            cplx* tmp = src; src = dst; dst = tmp;
        }
        // 3. split into the 257 bins of the real FFT, power
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            const cplx zk = src[k];
            cplx zm = src[(256 - k) & 255];
            zm.y = -zm.y;
            const cplx xe = {0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y)};
            const cplx dd = csub(zk, zm);
            const cplx xo = {0.5f * dd.y, -0.5f * dd.x};    // (zk - zm) / (2i)
            const cplx X = cadd(xe, cmul(tw512[k], xo));
            pw[k] = X.x * X.x + X.y * X.y;
        }
        // bin 256; bins 257..263 are read (with zero weight) by the mel loop and must be finite
        if (lane < 8) { const cplx z0 = src[0]; const float r = z0.x - z0.y; pw[256 + lane] = lane ? 0.f : r * r; }
        WAVE_SYNC();
        // 4. mel triangles + log: fixed-trip loop over the dense weight table (zeros beyond a triangle; the
        //    bins read there are finite: pw[257..263] is kept zero), addresses are base + immediates
        {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < MAXC; ++q) s += pw[mst + q] * tabs[TB_MWD + q * NMEL + ml];
            if (lane < NMEL) lm[lane] = logf(s + 1e-6f);
        }
        WAVE_SYNC();
        // 5. DCT-II (ortho) and store: this lane's column of the matrix is in registers
        {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < NMEL; ++n) s += lm[n] * dct[n];
            if (lane < NMFCC) out[((size_t)clip * out_frames + t) * NMFCC + lane] = s;
        }
        WAVE_SYNC();
    }
}
}  // namespace

int mfcc_build_tables(var_ctx* c) {
    std::vector<float> tb(TB_TOTAL, 0.f);
    int* it = (int*)tb.data();
    for (int i = 0; i < WIN; i++) tb[TB_WIN + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / WIN));
    for (int m = 0; m < 256; m++) {
        tb[TB_TW256 + 2 * m] = (float)cos(-2.0 * M_PI * m / 256.0);
        tb[TB_TW256 + 2 * m + 1] = (float)sin(-2.0 * M_PI * m / 256.0);
        tb[TB_TW512 + 2 * m] = (float)cos(-2.0 * M_PI * m / 512.0);
        tb[TB_TW512 + 2 * m + 1] = (float)sin(-2.0 * M_PI * m / 512.0);
    }
    for (int n = 0; n < NMEL; n++)
        for (int k = 0; k < NMFCC; k++) {
            double v = cos(M_PI / NMEL * (n + 0.5) * k) * sqrt(2.0 / NMEL);
            if (k == 0) v *= 1.0 / sqrt(2.0);
            tb[TB_DCT + n * NMFCC + k] = (float)v;
        }
    // HTK mel triangles: torchaudio.functional.melscale_fbanks(257, 0, 8000, 40, 16000, None, 'htk')
    const double sr = 16000.0;
    const double m_min = 0.0, m_max = 2595.0 * log10(1.0 + (sr / 2.0) / 700.0);
    double fpts[NMEL + 2];
    for (int i = 0; i < NMEL + 2; i++) {
        const double m = m_min + (m_max - m_min) * i / (NMEL + 1);
        fpts[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0);
    }
    int wo = 0;
    for (int m = 0; m < NMEL; m++) {
        int start = -1, count = 0;
        for (int k = 0; k < NFREQ; k++) {
            const double f = (sr / 2.0) * k / (NFREQ - 1);
            const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
            const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
            const double w = fmax(0.0, fmin(down, up));
            if (w > 0.0) {
                if (start < 0) start = k;
                if (k != start + count) { VAR_SET_ERR(c, "mfcc tables: non-contiguous mel filter"); return VAR_ERR_ARG; }
                if (wo + count >= 640) { VAR_SET_ERR(c, "mfcc tables: weight overflow"); return VAR_ERR_ARG; }
                tb[TB_MW + wo + count] = (float)w;
                count++;
            }
        }
        it[TB_MSTART + m] = start < 0 ? 0 : start;
        it[TB_MCOUNT + m] = count;
        it[TB_MOFF + m] = wo;
        wo += count;
    }
    for (int m = 0; m < NMEL; m++) {
        if (it[TB_MCOUNT + m] > MAXC || it[TB_MSTART + m] + MAXC - 1 > 263) { VAR_SET_ERR(c, "mfcc tables: triangle too wide"); return VAR_ERR_ARG; }
        for (int q = 0; q < it[TB_MCOUNT + m]; q++) tb[TB_MWD + q * NMEL + m] = tb[TB_MW + it[TB_MOFF + m] + q];
    }
    for (int i = 0; i < WIN; i++) tb[TB_WIN512 + WOFF + i] = tb[TB_WIN + i] * (1.f / 32768.f);   // exact: power of two
    static_assert(TB_TOTAL % 4 == 0, "table is copied as float4");
    VAR_HIP_CHECK(c, hipMalloc((void**)&c->mfcc_tab, sizeof(float) * TB_TOTAL));
    VAR_HIP_CHECK(c, hipMemcpy(c->mfcc_tab, tb.data(), sizeof(float) * TB_TOTAL, hipMemcpyHostToDevice));
    return VAR_OK;
}

int launch_mfcc(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                int pcm_stride, int out_frames, float* out) {
    ProfScope prof(c, s, TAG_MFCC);
    hipLaunchKernelGGL(mfcc_kernel, dim3(nclips), dim3(MW * 64), 0, s, pcm, lens, clip_index, pcm_stride, out_frames,
                       c->mfcc_tab, out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
