// torchaudio MFCC front-end for ANY (n_fft, win_length, hop_length) of Envs/audioLoader.py:23-31 -- the reference picks the
// STFT parameters by the dataset a clip comes from: 512 / 400 / 160 for GoogleCommand, FSC, ESC50 (the tuned kernel of
// mfcc.hip, which var_mfcc_ex dispatches to) and 1024 / 800 / 640 for NSynth and UrbanSound (this file).  Same algorithm
// as mfcc.hip (Envs/audioLoader.py:147-157 + :241-252): x / 32768 -> reflect-padded frames x periodic Hamming(win)
// centred in n_fft -> |rFFT|^2 -> 40 HTK mel triangles -> log(. + 1e-6) -> orthonormal DCT-II, T = 1 + N / hop frames.
// Not speed-tuned: one wavefront per frame, in-place radix-2 FFT of the zero-imaginary frame in the wave's LDS region,
// per-lane loops for the mel triangles and the DCT; the reference's Kuka / iTHOR configurations never take this path.
#include <math.h>

#include <map>
#include <tuple>
#include <vector>

#include "var_common.h"

namespace {
constexpr int NMEL = 40, NMFCC = 40;
constexpr int MAXFFT = 2048, WAVES = 4;

struct AnyTab {          // device table of one (n_fft, win, hop) configuration
    float* dev = nullptr;
    int n_fft = 0, log2n = 0, o_win = 0, o_tw = 0, o_dct = 0, o_mstart = 0, o_mcount = 0, o_mel = 0;
};
std::map<std::tuple<var_ctx*, int, int, int>, AnyTab>& tabs() { static std::map<std::tuple<var_ctx*, int, int, int>, AnyTab> t; return t; }

struct cpx { float x, y; };

__global__ void __launch_bounds__(WAVES * 64)
mfcc_any_kernel(const int16_t* __restrict__ pcm, const int* __restrict__ lens, const int* __restrict__ clip_index,
                int pcm_stride, int out_frames, int total_frames, int n_fft, int log2n, int hop,
                const float* __restrict__ tab, int o_win, int o_tw, int o_dct, int o_mstart, int o_mcount, int o_mel,
                float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cpx* z = (cpx*)lds + (size_t)wave * n_fft;                 // this wave's frame (n_fft complex points)
    float* lm = lds + (size_t)WAVES * n_fft * 2 + wave * 64;    // its 40 log-mel values
    const cpx* tw = (const cpx*)(tab + o_tw);
    const int* mstart = (const int*)(tab + o_mstart);
    const int* mcount = (const int*)(tab + o_mcount);
    const int nfreq = n_fft / 2 + 1;
    for (int f = blockIdx.x * WAVES + wave; f < total_frames; f += gridDim.x * WAVES) {
        const int clip = f / out_frames, t = f - clip * out_frames;
        int N = lens[clip];
        N = N < pcm_stride ? N : pcm_stride;
        float* dst = out + (size_t)f * NMFCC;
        if (N <= 0 || t >= 1 + N / hop) {                       // "empty" class / MFCC-domain zero padding
            if (lane < NMFCC) dst[lane] = 0.f;
            continue;
        }
        const int16_t* sig = pcm + (size_t)(clip_index ? clip_index[clip] : clip) * pcm_stride;
        const int p0 = t * hop - n_fft / 2;
        // windowed frame in bit-reversed order (decimation in time)
        for (int i = lane; i < n_fft; i += 64) {
            const float w = tab[o_win + i];
            float v = 0.f;
            if (w != 0.f) {
                int pos = p0 + i;                                // reflect padding of the centred STFT
                if (pos < 0) pos = -pos;
                if (pos >= N) pos = 2 * (N - 1) - pos;
                pos = pos < 0 ? 0 : (pos >= N ? N - 1 : pos);
                v = (float)sig[pos] * w;
            }
            z[__brev((unsigned)i) >> (32 - log2n)] = cpx{v, 0.f};
        }
        __builtin_amdgcn_wave_barrier();
        for (int s = 0; s < log2n; ++s) {
            const int hs = 1 << s;
            for (int b = lane; b < n_fft / 2; b += 64) {
                const int j = b & (hs - 1), i0 = ((b >> s) << (s + 1)) + j, i1 = i0 + hs;
                const cpx w = tw[j << (log2n - 1 - s)];          // exp(-2 pi i j / 2^(s+1))
                const cpx a = z[i0], c = z[i1];
                const cpx m = {c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x};
                z[i0] = cpx{a.x + m.x, a.y + m.y};
                z[i1] = cpx{a.x - m.x, a.y - m.y};
            }
            __builtin_amdgcn_wave_barrier();
        }
        // power spectrum in place (real part of the first n_fft/2 + 1 points)
        for (int k = lane; k < nfreq; k += 64) { const cpx v = z[k]; ((float*)z)[2 * k] = v.x * v.x + v.y * v.y; }
        __builtin_amdgcn_wave_barrier();
        if (lane < NMEL) {
            float s = 0.f;
            const int k0 = mstart[lane], n = mcount[lane];
            for (int q = 0; q < n; ++q) s += ((const float*)z)[2 * (k0 + q)] * tab[o_mel + (k0 + q) * NMEL + lane];
            lm[lane] = logf(s + 1e-6f);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < NMFCC) {
            float s = 0.f;
            for (int n = 0; n < NMEL; ++n) s += lm[n] * tab[o_dct + n * NMFCC + lane];
            dst[lane] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int build_any(var_ctx* c, int n_fft, int win, int hop, AnyTab& T) {
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    const int nfreq = n_fft / 2 + 1;
    T.n_fft = n_fft; T.log2n = log2n;
    int o = 0;
    T.o_win = o; o += n_fft;
    T.o_tw = o; o += n_fft;                 // n_fft/2 complex
    T.o_dct = o; o += NMEL * NMFCC;
    T.o_mstart = o; o += NMEL;
    T.o_mcount = o; o += NMEL;
    T.o_mel = o; o += nfreq * NMEL;
    std::vector<float> tb(o, 0.f);
    int* it = (int*)tb.data();
    const int left = (n_fft - win) / 2;
    for (int i = 0; i < win; i++) tb[T.o_win + left + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / win)) * (1.f / 32768.f);
    for (int k = 0; k < n_fft / 2; k++) {
        tb[T.o_tw + 2 * k] = (float)cos(-2.0 * M_PI * k / n_fft);
        tb[T.o_tw + 2 * k + 1] = (float)sin(-2.0 * M_PI * k / n_fft);
    }
    for (int n = 0; n < NMEL; n++)
        for (int k = 0; k < NMFCC; k++) {
            double v = cos(M_PI / NMEL * (n + 0.5) * k) * sqrt(2.0 / NMEL);
            if (k == 0) v *= 1.0 / sqrt(2.0);
            tb[T.o_dct + n * NMFCC + k] = (float)v;
        }
    const double sr = 16000.0, m_max = 2595.0 * log10(1.0 + (sr / 2.0) / 700.0);
    double fpts[NMEL + 2];
    for (int i = 0; i < NMEL + 2; i++) fpts[i] = 700.0 * (pow(10.0, (m_max * i / (NMEL + 1)) / 2595.0) - 1.0);
    for (int m = 0; m < NMEL; m++) {
        int start = -1, last = -1;
        for (int k = 0; k < nfreq; k++) {
            const double f = (sr / 2.0) * k / (nfreq - 1);
            const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
            const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
            const double w = fmax(0.0, fmin(down, up));
            tb[T.o_mel + k * NMEL + m] = (float)w;
            if (w > 0.0) { if (start < 0) start = k; last = k; }
        }
        it[T.o_mstart + m] = start < 0 ? 0 : start;
        it[T.o_mcount + m] = start < 0 ? 0 : last - start + 1;
    }
    VAR_HIP_CHECK(c, hipMalloc((void**)&T.dev, sizeof(float) * o));
    VAR_HIP_CHECK(c, hipMemcpy(T.dev, tb.data(), sizeof(float) * o, hipMemcpyHostToDevice));
    return VAR_OK;
}
}  // namespace

// (tables are built on the first call for a configuration -- the only allocating call: make it once outside graph
// capture -- and live until var_destroy)
int launch_mfcc_any(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                    int pcm_stride, int out_frames, int n_fft, int win, int hop, float* out) {
    if (n_fft < 64 || n_fft > MAXFFT || (n_fft & (n_fft - 1)) || win < 1 || win > n_fft || hop < 1) {
        VAR_SET_ERR(c, "var_mfcc_ex: n_fft %d (a power of two in 64..%d), win_length %d (<= n_fft), hop_length %d", n_fft, MAXFFT, win, hop);
        return VAR_ERR_ARG;
    }
    AnyTab& T = tabs()[std::make_tuple(c, n_fft, win, hop)];
    if (!T.dev) {
        int rc = build_any(c, n_fft, win, hop, T);
        if (rc != VAR_OK) return rc;
        if ((rc = retire_block(c, T.dev)) != VAR_OK) return rc;          // freed by var_destroy
    }
    const long total = (long)nclips * out_frames;
    const int lds_bytes = (WAVES * n_fft * 2 + WAVES * 64) * 4;
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)mfcc_any_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (WAVES * MAXFFT * 2 + WAVES * 64) * 4));
        attr_set |= var_dev_bit(c);
    }
    long blocks = (total + WAVES - 1) / WAVES;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mfcc_any_kernel, dim3((int)blocks), dim3(WAVES * 64), lds_bytes, s, pcm, lens, clip_index, pcm_stride,
                       out_frames, (int)total, n_fft, T.log2n, hop, T.dev, T.o_win, T.o_tw, T.o_dct, T.o_mstart, T.o_mcount,
                       T.o_mel, out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

void mfcc_any_forget(var_ctx* c) {              // var_destroy: the device blocks are freed with the retired list
    auto& m = tabs();
    for (auto it = m.begin(); it != m.end();) { if (std::get<0>(it->first) == c) it = m.erase(it); else ++it; }
}
