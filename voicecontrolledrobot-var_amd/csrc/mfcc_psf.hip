// The iTHOR/FSC audio front-end on the GPU: python_speech_features.mfcc as the reference calls it
// (Envs/audioLoader.py:158-161: winlen .025, winstep .01, numcep 40, nfilt 40, nfft 512, winfunc np.hamming, library
// defaults preemph .97, ceplifter 22, appendEnergy True; the int16 signal is NOT normalised) followed by
// processSoundFeat (Envs/audioLoader.py:241-252):
//   y[n] = x[n] - .97 x[n-1] -> frames of 400 every 160 (zero-padded tail, T = 1 + ceil((N-400)/160)) x symmetric
//   Hamming(400) -> |rFFT_512|^2 / 512 -> 40 triangles on bins floor(513 f / 16000) -> log -> orthonormal DCT-II ->
//   lifter 1 + 11 sin(pi k / 22) -> coefficient 0 := log(frame energy); rows beyond T zero, beyond out_frames dropped.
// The kernel is mfcc.hip's tile kernel (16 lanes per frame, packed-f32 radix-16 x 16 FFT, mel / DCT on the matrix cores) in
// its PSF flavour; the library computes in float64, the kernel in float32 (tests/test_gpu_ithor.py::test_psf_mfcc_vs_oracle).
#include "var_common.h"

extern "C" int var_mfcc_psf(var_ctx* c, void* stream, const int16_t* pcm, const int* lens, const int* clip_index,
                            int nclips, int pcm_stride, int out_frames, float* out) {
    if (!c) return VAR_ERR_ARG;
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    if (!pcm || !lens || !out || nclips < 1 || out_frames < 1 || pcm_stride < 1 || (pcm_stride & 1)) {
        VAR_SET_ERR(c, "var_mfcc_psf: bad argument (pcm_stride must be even)");
        return VAR_ERR_ARG;
    }
    return launch_mfcc_psf(c, (hipStream_t)stream, pcm, lens, clip_index, nclips, pcm_stride, out_frames, out);
}
