// The iTHOR/FSC audio front-end on the GPU: python_speech_features.mfcc as the reference calls it
// (Envs/audioLoader.py:158-161: winlen .025, winstep .01, numcep 40, nfilt 40, nfft 512, winfunc np.hamming, library
// defaults preemph .97, ceplifter 22, appendEnergy True; the int16 signal is NOT normalised) followed by
// processSoundFeat (Envs/audioLoader.py:241-252):
//   y[n] = x[n] - .97 x[n-1] -> frames of 400 every 160 (zero-padded tail, T = 1 + ceil((N-400)/160)) x symmetric
//   Hamming(400) -> |rFFT_512|^2 / 512 -> 40 triangles on bins floor(513 f / 16000) -> log -> orthonormal DCT-II ->
//   lifter 1 + 11 sin(pi k / 22) -> coefficient 0 := log(frame energy); rows beyond T zero, beyond out_frames dropped.
// Same wave-per-frame radix-4 Stockham FFT as mfcc.hip; the library computes in float64, this kernel in float32.
#include <math.h>

#include <vector>

#include "var_common.h"

namespace {
constexpr int NFFT = 512, WIN = 400, HOP = 160, NMEL = 40, NMFCC = 40;
constexpr int MAXC = 36;
constexpr int TB_TW256 = 0;               // 256 x (cos, sin)
constexpr int TB_TW512 = 512;
constexpr int TB_DCT = 1024;              // [n][k] 40 x 40, lifter folded in
constexpr int TB_MSTART = TB_DCT + 1600;  // 40 ints
constexpr int TB_MWD = TB_MSTART + 40;    // dense [q][40]
constexpr int TB_WIN = TB_MWD + MAXC * 40;  // 400
constexpr int TB_TOTAL = TB_WIN + 400;
static_assert(TB_TOTAL % 4 == 0, "table is copied as float4");

struct cplx { float x, y; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()

constexpr int MW = 16;
constexpr float kEps = 2.220446049250313e-16f;    // numpy.finfo(float).eps, the library's stand-in for log(0)

__global__ void __launch_bounds__(MW * 64)
mfcc_psf_kernel(const int16_t* __restrict__ pcm, const int* __restrict__ lens, const int* __restrict__ clip_index,
                int pcm_stride, int out_frames, const float* __restrict__ tab, float* __restrict__ out) {
    __shared__ float tabs[TB_TOTAL];
    __shared__ cplx bufA[MW][256];
    __shared__ cplx bufB[MW][256 + 24];              // power spectrum (257 + zero tail up to 300) + log-mel (40)
    const int clip = blockIdx.x, chunk = blockIdx.y, nchunk = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < TB_TOTAL / 4; e += MW * 64) ((float4*)tabs)[e] = ((const float4*)tab)[e];
    const int N = lens[clip];
    const int T = N > WIN ? 1 + (N - WIN + HOP - 1) / HOP : 1;
    const int16_t* sig = pcm + (size_t)(clip_index ? clip_index[clip] : clip) * pcm_stride;
    const int* itab = (const int*)tabs;
    const cplx* tw256 = (const cplx*)(tabs + TB_TW256);
    const cplx* tw512 = (const cplx*)(tabs + TB_TW512);
    float* xs = (float*)bufA[wave];
    __syncthreads();
    const int ml = lane < NMEL ? lane : 0;
    const int mst = itab[TB_MSTART + ml];
    float dct[NMEL];
#pragma unroll
    for (int n = 0; n < NMEL; ++n) dct[n] = tabs[TB_DCT + n * NMFCC + ml];
    float* pw = (float*)bufB[wave];
    float* lm = pw + 304;
    // frames of this clip are dealt to (chunk, wave): long clips (600 frames) spread over several workgroups
    for (int t = chunk * MW + wave; t < out_frames; t += MW * nchunk) {
        const bool live = t < T && N > 0;
        if (!live) {
            if (lane < NMFCC) out[((size_t)clip * out_frames + t) * NMFCC + lane] = 0.f;
            continue;
        }
        const int p0 = t * HOP;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + 64 * i;
            float v = 0.f;
            const int pos = p0 + idx;
            if (idx < WIN && pos < N) {
                const float cur = (float)sig[pos];
                const float prev = pos > 0 ? (float)sig[pos - 1] : 0.f;
                v = (pos > 0 ? cur - 0.97f * prev : cur) * tabs[TB_WIN + idx];
            }
            xs[idx] = v;
        }
        WAVE_SYNC();
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
            const cplx a3 = {d.y, -d.x};
            const int idx = (lane / Ns) * Ns * 4 + jm;
            dst[idx] = cadd(a0, a2);
            dst[idx + Ns] = cadd(a1, a3);
            dst[idx + 2 * Ns] = csub(a0, a2);
            dst[idx + 3 * Ns] = csub(a1, a3);
            WAVE_SYNC();
            cplx* tmp = src; src = dst; dst = tmp;
        }
        // 257 bins of the real FFT, power / 512, frame energy
        float e = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            const cplx zk = src[k];
            cplx zm = src[(256 - k) & 255];
            zm.y = -zm.y;
            const cplx xe = {0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y)};
            const cplx dd = csub(zk, zm);
            const cplx xo = {0.5f * dd.y, -0.5f * dd.x};
            const cplx X = cadd(xe, cmul(tw512[k], xo));
            const float p = (X.x * X.x + X.y * X.y) * (1.f / 512.f);
            pw[k] = p;
            e += p;
        }
        if (lane < 44) {
            const cplx z0 = src[0]; const float r = z0.x - z0.y;
            const float p = lane ? 0.f : r * r * (1.f / 512.f);
            pw[256 + lane] = p;                       // bin 256, zeros up to 299
            e += p;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (e == 0.f) e = kEps;
        WAVE_SYNC();
        {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < MAXC; ++q) s += pw[mst + q] * tabs[TB_MWD + q * NMEL + ml];
            if (s == 0.f) s = kEps;
            if (lane < NMEL) lm[lane] = logf(s);
        }
        WAVE_SYNC();
        {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < NMEL; ++n) s += lm[n] * dct[n];
            if (lane == 0) s = logf(e);               // appendEnergy
            if (lane < NMFCC) out[((size_t)clip * out_frames + t) * NMFCC + lane] = s;
        }
        WAVE_SYNC();
    }
}
}  // namespace

static int mfcc_psf_tables(var_ctx* c) {
    if (c->mfcc_psf_tab) return VAR_OK;
    std::vector<float> tb(TB_TOTAL, 0.f);
    int* it = (int*)tb.data();
    for (int m = 0; m < 256; m++) {
        tb[TB_TW256 + 2 * m] = (float)cos(-2.0 * M_PI * m / 256.0);
        tb[TB_TW256 + 2 * m + 1] = (float)sin(-2.0 * M_PI * m / 256.0);
        tb[TB_TW512 + 2 * m] = (float)cos(-2.0 * M_PI * m / 512.0);
        tb[TB_TW512 + 2 * m + 1] = (float)sin(-2.0 * M_PI * m / 512.0);
    }
    for (int i = 0; i < WIN; i++) tb[TB_WIN + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / (WIN - 1)));   // np.hamming
    for (int n = 0; n < NMEL; n++)
        for (int k = 0; k < NMFCC; k++) {
            double v = cos(M_PI / NMEL * (n + 0.5) * k) * sqrt(2.0 / NMEL);
            if (k == 0) v *= 1.0 / sqrt(2.0);
            v *= 1.0 + (22.0 / 2.0) * sin(M_PI * k / 22.0);          // lifter(cepstra, L=22)
            tb[TB_DCT + n * NMFCC + k] = (float)v;
        }
    // get_filterbanks(40, 512, 16000, 0, 8000): triangles between bins floor((nfft+1) * hz / samplerate)
    const double lowmel = 0.0, highmel = 2595.0 * log10(1.0 + 8000.0 / 700.0);
    double bin[NMEL + 2];
    for (int i = 0; i < NMEL + 2; i++) {
        const double mel = lowmel + (highmel - lowmel) * i / (NMEL + 1);
        bin[i] = floor((NFFT + 1) * (700.0 * (pow(10.0, mel / 2595.0) - 1.0)) / 16000.0);
    }
    for (int j = 0; j < NMEL; j++) {
        const int b0 = (int)bin[j], b1 = (int)bin[j + 1], b2 = (int)bin[j + 2];
        if (b2 - b0 > MAXC || b0 + MAXC - 1 > 299) { VAR_SET_ERR(c, "mfcc_psf tables: triangle too wide"); return VAR_ERR_ARG; }
        it[TB_MSTART + j] = b0;
        for (int i = b0; i < b1; i++) tb[TB_MWD + (i - b0) * NMEL + j] = (float)((i - bin[j]) / (bin[j + 1] - bin[j]));
        for (int i = b1; i < b2; i++) tb[TB_MWD + (i - b0) * NMEL + j] = (float)((bin[j + 2] - i) / (bin[j + 2] - bin[j + 1]));
    }
    VAR_HIP_CHECK(c, hipMalloc((void**)&c->mfcc_psf_tab, sizeof(float) * TB_TOTAL));
    VAR_HIP_CHECK(c, hipMemcpy(c->mfcc_psf_tab, tb.data(), sizeof(float) * TB_TOTAL, hipMemcpyHostToDevice));
    return VAR_OK;
}

extern "C" int var_mfcc_psf(var_ctx* c, void* stream, const int16_t* pcm, const int* lens, const int* clip_index,
                            int nclips, int pcm_stride, int out_frames, float* out) {
    if (!c) return VAR_ERR_ARG;
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    if (!pcm || !lens || !out || nclips < 1 || out_frames < 1 || pcm_stride < 1 || (pcm_stride & 1)) {
        VAR_SET_ERR(c, "var_mfcc_psf: bad argument (pcm_stride must be even)");
        return VAR_ERR_ARG;
    }
    int r = mfcc_psf_tables(c);
    if (r != VAR_OK) return r;
    const int nchunk = (out_frames + 8 * MW - 1) / (8 * MW);      // about 8 frames per wave
    hipLaunchKernelGGL(mfcc_psf_kernel, dim3(nclips, nchunk), dim3(MW * 64), 0, (hipStream_t)stream, pcm, lens, clip_index,
                       pcm_stride, out_frames, c->mfcc_psf_tab, out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
