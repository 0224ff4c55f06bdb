// Audio front-end on the GPU: int16 PCM -> (frames, 40) MFCC, the work the reference does per
// item inside its DataLoader workers with torchaudio.transforms.MFCC
// (Envs/audioLoader.py:147-157) followed by processSoundFeat (Envs/audioLoader.py:241-252):
//   x/32768 -> reflect-padded frames (n_fft 512, hop 160) x periodic Hamming(400) centred in 512
//   -> |rFFT|^2 -> 40 HTK-mel triangles -> log(. + 1e-6) -> orthonormal DCT-II -> (T, 40),
//   T = 1 + N/160; rows beyond T are zero, rows beyond out_frames are dropped.
//
// Work unit = a TILE of 16 consecutive output frames of the flat (clip, frame) list, one 4-wave workgroup per tile
// (persistent, two workgroups per CU):
//   * FFT on the vector ALU, 16 LANES PER FRAME (4 frames per wave): the 512-point real FFT is a 256-point complex FFT of
//     z[n] = x[2n] + i x[2n+1], factored 256 = 16 x 16 -- every lane runs a 16-point FFT on registers (two radix-4
//     levels), multiplies by the W256 twiddles it keeps in registers, the 16 x 16 transpose goes once through a padded
//     LDS tile, a second 16-point FFT, then the even/odd split to the 257 power bins.  Two LDS exchanges per frame instead
//     of the four of a radix-4 Stockham FFT, a quarter of its wave-instructions.
//   * mel (257 -> 40) and DCT (40 -> 40) on the MATRIX cores over the tile's 16 frames (v_mfma_f32_16x16x4_f32): the mel
//     triangles of 16 consecutive filters only touch a band of bins (12 / 28 / 32 k-steps for the three filter tiles), the
//     B operands (filter weights, DCT columns) are wave constants held in registers, the A operand is the power tile in
//     LDS with a row pitch of 258 floats (conflict-free for the 16 x 4 operand read).
// PCM is read as 4-byte sample pairs straight from HBM/L2 (each sample is touched by 3.2 frames); frames that overlap
// either end of the clip (reflect padding) take a per-sample path.  Tables are computed once on the host in double.
#include <math.h>

#include <vector>

#include "var_common.h"

namespace {
constexpr int NFFT = 512, WIN = 400, HOP = 160, NMEL = 40, NMFCC = 40, NFREQ = 257;
constexpr int WOFF = (NFFT - WIN) / 2;   // 56

// mel filter tiles: filters [16 t, 16 t + 16) read bins [kMelK0[t], kMelK0[t] + 4 kMelSteps[t])
constexpr int kMelK0[3] = {0, 36, 132};
constexpr int kMelSteps[3] = {12, 28, 32};            // multiples of 4: every wave takes the same number of k-steps of a tile
constexpr int kMelSlots = (12 + 28 + 32) / 4;         // 18 B-operand registers per wave

// table layout (floats)
constexpr int TB_WIN = 0;                 // 400 (host-side only)
constexpr int TB_TW256 = 400;             // 256 x (cos, sin) of -2 pi m / 256
constexpr int TB_TW512 = TB_TW256 + 512;  // 256 x (cos, sin) of -2 pi k / 512
constexpr int TB_WIN512 = TB_TW512 + 512; // 512: Hamming(400) / 32768 centred in the frame, 0 outside
constexpr int TB_MELB = TB_WIN512 + 512;  // [wave 4][slot 18][lane 64]: B operand of the mel products
constexpr int TB_DCTB = TB_MELB + 4 * kMelSlots * 64;   // [tile 3][step 10][lane 64]: B operand of the DCT products
constexpr int TB_TOTAL = TB_DCTB + 3 * 10 * 64;          // a multiple of 4

// Complex numbers are float pairs in even-aligned register pairs so that the packed-f32 instructions of gfx950
// (v_pk_add/mul/fma_f32: two f32 operations per lane per issue) carry the FFT; the swaps and sign flips of complex
// arithmetic ride on the instructions' op_sel / neg modifiers (hipcc emits separate v_xor + v_mov for them).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f2 cplx;

// a * w  (w = (cos, sin) in a VGPR pair)
__device__ __forceinline__ f2 cmul(f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                     // (a.x w.x, a.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"              // (-a.y w.y + t.x, a.x w.y + t.y)
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * (c + i s) with compile-time constants (the compiler keeps them in scalar registers)
__device__ __forceinline__ f2 cmulc(f2 a, float c, float s) {
    const f2 sw = {-a.y, a.x};
    return __builtin_elementwise_fma(sw, f2{s, s}, a * f2{c, c});
}
// a + (-i) t = (a.x + t.y, a.y - t.x);  a - (-i) t = (a.x - t.y, a.y + t.x)
__device__ __forceinline__ f2 add_mi(f2 a, f2 t) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}
__device__ __forceinline__ f2 sub_mi(f2 a, f2 t) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}

// 4-point DFT in place: (a, b, c, d) -> (X0, X1, X2, X3), forward transform (W4 = -i): 8 packed adds
__device__ __forceinline__ void dft4(cplx& a, cplx& b, cplx& c, cplx& d) {
    const cplx s0 = a + c, s1 = a - c, s2 = b + d, t = b - d;
    a = s0 + s2; b = add_mi(s1, t); c = s0 - s2; d = sub_mi(s1, t);
}

// 16-point forward DFT on registers: v[n] -> v[k].  n = 4 na + q, k = r + 4 s:
//   a[q][r] = W16^(q r) * DFT4_na(v[4 na + q])[r];   X[r + 4 s] = DFT4_q(a[q][r])[s]
__device__ __forceinline__ void fft16(cplx (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(v[q], v[q + 4], v[q + 8], v[q + 12]);       // v[q + 4 r] = a[q][r] before twiddling
    // W16^(q r), q, r in 1..3: exponents 1 2 3 / 2 4 6 / 3 6 9
    v[1 + 4] = cmulc(v[1 + 4], C1, -S1);
    v[1 + 8] = cmulc(v[1 + 8], H, -H);
    v[1 + 12] = cmulc(v[1 + 12], S1, -C1);
    v[2 + 4] = cmulc(v[2 + 4], H, -H);
    v[2 + 8] = cplx{v[2 + 8].y, -v[2 + 8].x};                                    // * (-i)
    v[2 + 12] = cmulc(v[2 + 12], -H, -H);
    v[3 + 4] = cmulc(v[3 + 4], S1, -C1);
    v[3 + 8] = cmulc(v[3 + 8], -H, -H);
    v[3 + 12] = cmulc(v[3 + 12], -C1, S1);
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);   // -> X[r + 4 s] at v[4 r + s]
    // natural order: X[k] for k = r + 4 s sits at v[4 r + s]: transpose the 4 x 4 index (register renaming)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s2 = r + 1; s2 < 4; ++s2) { const cplx t = v[4 * r + s2]; v[4 * r + s2] = v[4 * s2 + r]; v[4 * s2 + r] = t; }
}

// Waves own their four frames through the FFT: within a wave, DS operations execute in order, so a compiler-level
// wave barrier is all that separates the two sides of an exchange through the wave's private LDS region.
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()

constexpr int FT = 16;                    // frames per tile
constexpr int NTHR = 256;                 // 4 waves
// Round 4: THREE workgroups per CU instead of two (the kernel is bound by dependent-instruction latency, not by any pipe: a
// third resident workgroup is worth what it adds in waves).  Two budgets had to come down for that:
//   LDS 67 -> 44 KB: the power tile and the mel partial sums live INSIDE the transpose buffer -- a frame's region there holds
//     272 complex numbers during the FFT and is dead from the even/odd split to the next tile's FFT: its first 258 floats become
//     the frame's power row (written from registers behind a wave barrier, once the split has read its last partner bin), the
//     288 floats behind them a sixteenth of the partial-sum block;
//   VGPRs 244 -> <= 168: the two twiddle tables (64 registers of per-lane constants) are LDS operands now (one ds_read_b64 per
//     use, the four frames of a wave read the same addresses), and the next tile's samples are requested behind the split, when
//     the FFT's 32 data registers are dead, instead of before it.
constexpr int TPITCH = 273;               // complex elements per frame region (272 used by the 17 x 16 transpose tile): 546 floats == 2 mod 32
constexpr int PPITCH = 2 * TPITCH;        // floats between two frames' power rows: conflict-free for the 16 x 4 operand read
constexpr int PARTOFF = 272, PARTLEN = 192;   // a frame region's share of the partial-sum block: floats [272, 464)
constexpr int LPITCH = 42;                // floats per frame of the log-mel tile (10 i mod 32 distinct for i < 16)

// LDS carve-up (floats): 44 KB per workgroup incl. the static arrays
constexpr int L_WIN = 0, L_TW = 512, L_TW5 = 1024, L_TBUF = 1536, L_TOTAL = L_TBUF + 2 * FT * TPITCH;
constexpr int MFCC_LDS_BYTES = L_TOTAL * 4;     // dynamic part (40 KB)
static_assert(258 <= PARTOFF && PARTOFF + PARTLEN <= PPITCH && 16 * PARTLEN == 4 * 3 * 4 * 64, "power row and partial sums fit a frame's region");

typedef float f32x4m __attribute__((ext_vector_type(4)));

struct FrameRef { const int16_t* sig; int p0, N; bool live, interior; };
struct FrameMeta { int clip, t, N, row; bool inb; };   // the two dependent index loads (lens, clip_index), one tile earlier

PH_DECL();
#ifdef VAR_PHASES
extern "C" int var_debug_phases_mfcc(unsigned long long* out) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

// PSF = the python_speech_features flavour of the same pipeline (iTHOR / FSC; header of mfcc_psf.hip): frames start at
// t * 160 with a zero-padded tail instead of being centred with reflected ends, the signal is pre-emphasised
// (y[n] = x[n] - .97 x[n-1], y[0] = x[0]) and not normalised, symmetric Hamming, power / 512, the library's integer-bin
// triangles, log with eps for exact zeros, liftered DCT, coefficient 0 := log(frame energy).  What differs in the kernel:
// where a frame's samples come from, the pre-emphasis (the previous sample is the neighbouring lane's: one DPP move),
// and the energy (Parseval: sum_k<=256 |X_k|^2 = (512 sum_n f_n^2 + |X_0|^2 + |X_256|^2) / 2 -- the mel products only
// read bands of the spectrum); everything else is table contents.
constexpr float kPsfEps = 2.220446049250313e-16f;    // numpy.finfo(float).eps, the library's stand-in for log(0)
constexpr float kPreemph = 0.97f;

// lane j of each 16-lane row receives lane (j - 1) & 15's value (row_ror:1)
__device__ __forceinline__ uint32_t row_ror1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false);
}
__device__ __forceinline__ float row_sum16(float e) {          // every lane of a 16-lane row gets the row's sum
    e += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x128, 0xf, 0xf, false));
    e += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x124, 0xf, 0xf, false));
    e += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x122, 0xf, 0xf, false));
    e += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x121, 0xf, 0xf, false));
    return e;
}

template <bool PSF>
__global__ void __launch_bounds__(NTHR, 3)
mfcc_kernel(const int16_t* __restrict__ pcm, const int* __restrict__ lens, const int* __restrict__ clip_index,
            int pcm_stride, int out_frames, int total_frames, const float* __restrict__ tab, float* __restrict__ out) {
    // separate LDS objects, so that the compiler may move a wave's reads of one past its writes of another (with one
    // array the 16 partner-bin reads of the split were each held behind the previous bin's power store)
    extern __shared__ __attribute__((aligned(16))) float lds[];         // window + twiddle tables, transpose tile (dynamic part)
    __shared__ __attribute__((aligned(16))) float lmel[FT * LPITCH];    // log-mel tile (A operand of the DCT products)
    __shared__ int live_s[2 * FT];                                      // [parity][frame]
    __shared__ float en_s[2 * FT];                                      // PSF: frame energy
    const float* win = lds + L_WIN;                 // window / 32768, zero outside its 400 taps
    cplx* tbuf = (cplx*)(lds + L_TBUF);             // transpose tile; then the spectrum Z[k] (256 per frame); then power rows + partial sums
    float* pw = lds + L_TBUF;                       // power row of frame f: pw[f * PPITCH + k], k = 0..257 (4 |X|^2; column 257 = 0)
    const cplx* twl = (const cplx*)(lds + L_TW);    // W256^(j k1) at [k1][j]
    const cplx* tw5l = (const cplx*)(lds + L_TW5);  // W512^(j + 16 m) at [m][j]
    auto part_at = [&](int e) { return lds + L_TBUF + (e / PARTLEN) * PPITCH + PARTOFF + (e % PARTLEN); };   // partial-sum block, element e
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int fi = 4 * wave + g;                                           // this lane group's frame within the tile

    const int ntiles = (total_frames + FT - 1) / FT;
    cplx* tb = tbuf + fi * TPITCH;
    float* pf = pw + fi * PPITCH;

    const float inv_of = 1.f / (float)out_frames;
    auto meta_of = [&](int tile) {
        FrameMeta m;
        const int f = tile * FT + fi;
        m.inb = tile < ntiles && f < total_frames;
        // clip = f / out_frames through a float reciprocal (f < 2^23 frames is checked by the launcher) + one correction
        int clip = (int)((float)f * inv_of), t = f - clip * out_frames;
        if (t < 0) { --clip; t += out_frames; } else if (t >= out_frames) { ++clip; t -= out_frames; }
        if (!m.inb) { clip = 0; t = 0; }
        m.clip = clip; m.t = t;
        m.N = lens[clip];
        m.row = clip_index ? clip_index[clip] : clip;
        return m;
    };
    auto frame_of = [&](const FrameMeta& m) {
        FrameRef r;
        r.N = m.N < pcm_stride ? m.N : pcm_stride;
        r.sig = pcm + (size_t)m.row * pcm_stride;
        if (PSF) {
            const int T = r.N > WIN ? 1 + (r.N - WIN + HOP - 1) / HOP : 1;
            r.live = m.inb && r.N > 0 && m.t < T;
            r.p0 = m.t * HOP;
            r.interior = r.live && r.p0 > 0 && r.p0 + NFFT <= r.N && !(pcm_stride & 1);
        } else {
            r.live = m.inb && r.N > 0 && m.t < 1 + r.N / HOP;
            r.p0 = m.t * HOP - NFFT / 2;
            r.interior = r.live && r.p0 >= 0 && r.p0 + NFFT <= r.N && !(pcm_stride & 1);
        }
        return r;
    };
    // sample pairs of the frame, one tile ahead: lane j takes z[16 n1 + j] = (x[32 n1 + 2 j], x[32 n1 + 2 j + 1]).
    // Interior frames: one 4-byte load per pair.  Frames that overlap an end of the clip (reflect padding of the centred
    // STFT; t = 0, 1 and the last four of a full clip): two 2-byte loads at reflected positions -- the window table is zero
    // outside its 400 taps, so positions only have to be valid, not meaningful, there.
    auto fetch = [&](const FrameRef& r, uint32_t (&two)[16]) {
        if (__builtin_amdgcn_ballot_w64(r.live && !r.interior) == 0ull) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) two[n1] = r.live ? *(const uint32_t*)(r.sig + r.p0 + 32 * n1 + 2 * j) : 0u;
        } else {
            // all 32 loads first (branch-free: dead lanes read sample 0 of a valid row), then the packing -- written
            // with a per-pair `if (live)` hipcc waits for every pair of loads before issuing the next
            const int Nc = r.N > 0 ? r.N : 1;
            uint16_t lo[16], hi[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                int a = r.p0 + 32 * n1 + 2 * j, b = a + 1;
                if (!PSF) {       // reflected ends (PSF: positions past the end are masked after the pre-emphasis)
                    a = a < 0 ? -a : a; b = b < 0 ? -b : b;
                    a = a >= Nc ? 2 * (Nc - 1) - a : a; b = b >= Nc ? 2 * (Nc - 1) - b : b;
                }
                a = a < 0 ? 0 : (a >= Nc ? Nc - 1 : a); b = b < 0 ? 0 : (b >= Nc ? Nc - 1 : b);
                a = r.live ? a : 0; b = r.live ? b : 0;
                lo[n1] = (uint16_t)r.sig[a];
                hi[n1] = (uint16_t)r.sig[b];
            }
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) two[n1] = r.live ? ((uint32_t)lo[n1] | ((uint32_t)hi[n1] << 16)) : 0u;
        }
    };

    // PSF: the sample in front of the frame (0 for the clip's first frame: y[0] = x[0]), in the high half like a pair's
    auto fetch_prev = [&](const FrameRef& r) { return (PSF && r.live && r.p0 > 0) ? (uint32_t)(uint16_t)r.sig[r.p0 - 1] << 16 : 0u; };

    PH_INIT(5);
    int tile = blockIdx.x;
    FrameRef fr = frame_of(meta_of(tile));
    uint32_t two[16];
    fetch(fr, two);
    uint32_t pm1 = fetch_prev(fr);
    FrameMeta mnext = meta_of(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    // Tables AFTER the first tile's requests: a workgroup lives for only ~4 tiles (768 workgroups), and its start used to be
    // three dependent round trips in a row -- tables, then the clip lookups, then the samples (~12 K cycles, a sixth of its life).
    // Now the lookups and the samples are in flight while the tables are read and written to LDS.
    for (int e = tid; e < 512; e += NTHR) lds[L_WIN + e] = tab[TB_WIN512 + e];
    {   // twiddle tables: entry (k1 | m, j2) = tid
        const int a = tid >> 4, j2 = tid & 15;
        ((cplx*)(lds + L_TW))[tid] = ((const cplx*)(tab + TB_TW256))[(j2 * a) & 255];
        ((cplx*)(lds + L_TW5))[tid] = ((const cplx*)(tab + TB_TW512))[j2 + 16 * a];
    }
    if (tid < 2 * FT) live_s[tid] = 0;
    // mel / DCT B operands stay in registers (28); the twiddles are LDS operands (see the head of this file)
    float melb[kMelSlots], dctb[10];
#pragma unroll
    for (int s = 0; s < kMelSlots; ++s) melb[s] = tab[TB_MELB + (wave * kMelSlots + s) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 10; ++s) dctb[s] = tab[TB_DCTB + ((wave < 3 ? wave : 0) * 10 + s) * 64 + lane];
    __syncthreads();
    int par = 0;
    // (measured: handing the tiles out through a device-wide atomic counter -- to balance the runs of dead tiles of
    //  "empty" clips -- costs more than it balances: 3200 draws on one address take the kernel from 40 to 55 us)
    // (measured, round 4: dealing the tiles by LIVE rank -- every workgroup scans the clips' lengths, a prefix table in LDS, equal
    //  shares of live tiles, tiles cut at clip ends -- 44.4 us instead of 40.6: the launch is not waiting for unlucky workgroups,
    //  a CU with three resident workgroups simply turns out ~50 tiles per us however they are dealt; the 12 % of extra
    //  (quarter-filled) tiles and the scan cost what the balance gave.  Same finding as round 3's live-tiles-first pre-pass.)
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x, par ^= FT) {
        PH(0);
        if (j == 0) live_s[par + fi] = fr.live ? 1 : 0;
        // a tile without a single live frame (the "empty" class, dataset.py:37-38, covers whole clips; so does the
        // zero padding of short clips) is all zeros: skip the transform.  The vote is workgroup-uniform.
        if (__syncthreads_or(fr.live) == 0) {
            for (int e = tid; e < FT * NMFCC; e += NTHR) {
                const long o = (long)tile * FT * NMFCC + e;
                if (o < (long)total_frames * NMFCC) out[o] = 0.f;
            }
            fr = frame_of(mnext);
            fetch(fr, two);
            pm1 = fetch_prev(fr);
            mnext = meta_of(tile + 2 * gridDim.x);
            continue;
        }
        PH(1);
        cplx v[16];
        // 1. windowed samples
        float en = 0.f;
        if (PSF) {
            const bool edge = __builtin_amdgcn_ballot_w64(fr.live && !fr.interior) != 0ull;      // wave-uniform
            uint32_t carry = pm1;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const uint32_t rot = row_ror1(two[n1]);              // the pair in front of this lane's (lane 0: of the next n1's)
                const uint32_t before = j == 0 ? carry : rot;
                carry = rot;
                const float xl = (float)(int16_t)(two[n1] & 0xffff), xh = (float)((int32_t)two[n1] >> 16);
                const float xp = (float)((int32_t)before >> 16);
                f2 y = f2{xl - kPreemph * xp, xh - kPreemph * xl};
                if (edge) {                                          // zero padding past the clip's end
                    const int pos = fr.p0 + 32 * n1 + 2 * j;
                    y.x = pos < fr.N ? y.x : 0.f;
                    y.y = pos + 1 < fr.N ? y.y : 0.f;
                }
                v[n1] = y * *(const f2*)(win + 32 * n1 + 2 * j);
                const f2 sq = v[n1] * v[n1];
                en += sq.x + sq.y;
            }
            en = row_sum16(en);
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                v[n1] = f2{(float)(int16_t)(two[n1] & 0xffff), (float)((int32_t)two[n1] >> 16)} * *(const f2*)(win + 32 * n1 + 2 * j);
        }
        PH(2);
        // 2. 16-point FFT over n1 (this lane is n2 = j), twiddle W256^(n2 k1), transpose through LDS, FFT over n2
        fft16(v);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], twl[k1 * 16 + j]);
        PH(3);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) tb[k1 * 17 + j] = v[k1];
        WAVE_SYNC();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) v[n2] = tb[j * 17 + n2];
        WAVE_SYNC();
        PH(4);
        fft16(v);                                                   // v[k2] = Z[j + 16 k2]
        PH(5);
        // 3. spectrum to LDS in natural order (the partner bin 256 - k lives in another lane)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) tb[16 * k2 + j] = v[k2];
        WAVE_SYNC();
        // 4. even/odd split of the real transform, power x 4 (the 1/4 is folded into the mel weights): bins k = j + 16 m
        //    2 X[k] = (Zk + conj Zm) + W512^k * (-i) (Zk - conj Zm),  m = 256 - k
        float pwr[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int k = j + 16 * m;
            const cplx zk = v[m], zm = tb[(256 - k) & 255], w5 = tw5l[m * 16 + j];
            f2 sm, df, u, X;
            asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(sm) : "v"(zk), "v"(zm));             // Zk + conj Zm
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(df) : "v"(zk), "v"(zm));             // Zk - conj Zm
            // W * (-i) df = (df.y, -df.x) W.x + (df.x, df.y) W.y
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(u) : "v"(df), "v"(w5), "v"(sm));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(X) : "v"(df), "v"(w5), "v"(u));
            const f2 q = X * X;
            pwr[m] = q.x + q.y;
        }
        const float r256 = v[0].x - v[0].y, p256 = 4.f * r256 * r256;              // bin 256 = (Re Z0 - Im Z0)^2
        // the next tile's samples: requested here, where the FFT's registers are dead; they fly during the mel / log / DCT phases
        // (their clip row / length were fetched a tile earlier)
        const float en_keep = en;
        fr = frame_of(mnext);
        fetch(fr, two);
        pm1 = fetch_prev(fr);
        mnext = meta_of(tile + 2 * gridDim.x);
        __builtin_amdgcn_sched_barrier(0);          // keep the loads HERE (hipcc otherwise sinks them towards their first use)
        WAVE_SYNC();                                // every partner-bin read of this wave's frames is done: the rows may be overwritten
#pragma unroll
        for (int m = 0; m < 16; ++m) pf[j + 16 * m] = pwr[m];
        if (j == 0) {
            pf[256] = p256;
            pf[257] = 0.f; pf[258] = 0.f; pf[259] = 0.f;      // (the last k-steps of the third filter tile read bins up to 259: zero weights, finite operands)
            if (PSF) en_s[par + fi] = 0.5f * en_keep + (pwr[0] + p256) * (1.f / 4096.f);   // sum of the 257 bins of |X|^2 / 512
        }
        PH(6);
        __syncthreads();                                            // (1) power tile complete
        PH(7);
        // 5. mel triangles on the matrix cores: D[frame][filter] over this wave's k-steps of each filter tile
        f32x4m acc[3];
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) acc[tl] = f32x4m{0.f, 0.f, 0.f, 0.f};
        {
            const float* pa = pw + (lane & 15) * PPITCH + (lane >> 4);
            int slot = 0;
#pragma unroll
            for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                for (int q = 0; q < kMelSteps[tl] / 4; ++q, ++slot)
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kMelK0[tl] + 4 * (4 * q + wave)], melb[slot], acc[tl], 0, 0, 0);
        }
#pragma unroll
        for (int tl = 0; tl < 3; ++tl)
#pragma unroll
            for (int r = 0; r < 4; ++r) *part_at(((wave * 3 + tl) * 4 + r) * 64 + lane) = acc[tl][r];
        PH(8);
        __syncthreads();                                            // (2) partial sums in place
        PH(9);
        // 6. fold the four partials in fixed order, log(. + 1e-6)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int r = tid >> 6, l2 = tid & 63;                  // (r, lane') of filter tile q
            float s = *part_at(((0 * 3 + q) * 4 + r) * 64 + l2);
            s += *part_at(((1 * 3 + q) * 4 + r) * 64 + l2);
            s += *part_at(((2 * 3 + q) * 4 + r) * 64 + l2);
            s += *part_at(((3 * 3 + q) * 4 + r) * 64 + l2);
            const int frame = 4 * (l2 >> 4) + r, mel = 16 * q + (l2 & 15);
            if (mel < NMEL) lmel[frame * LPITCH + mel] = PSF ? __logf(s == 0.f ? kPsfEps : s) : __logf(s + 1e-6f);   // v_log_f32: 1 ulp
        }
        PH(10);
        __syncthreads();                                            // (3) log-mel tile complete
        PH(11);
        // 7. orthonormal DCT-II on the matrix cores (coefficient tile = wave, 10 k-steps) and store
        if (wave < 3) {
            f32x4m d = f32x4m{0.f, 0.f, 0.f, 0.f};
            const float* la = lmel + (lane & 15) * LPITCH + (lane >> 4);
#pragma unroll
            for (int s = 0; s < 10; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(la[4 * s], dctb[s], d, 0, 0, 0);
            const int coef = 16 * wave + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int frame = 4 * (lane >> 4) + r;
                const int fo = tile * FT + frame;
                float val = d[r];
                if (PSF && coef == 0) { const float e = en_s[par + frame]; val = __logf(e == 0.f ? kPsfEps : e); }    // appendEnergy
                if (coef < NMFCC && fo < total_frames) out[(size_t)fo * NMFCC + coef] = live_s[par + frame] ? val : 0.f;
            }
        }
        PH(12);
        // no barrier here: the next tile touches tbuf (wave-private), pw only before its barrier 1 -- which every wave
        // reaches after leaving step 5 of this tile --, part / lmel after its barriers 1 / 2, and the other half of live_s
    }
}
}  // namespace

// window / twiddle / mel / DCT tables of one flavour (Kuka: torchaudio.transforms.MFCC; psf: python_speech_features.mfcc)
static int build_tables(var_ctx* c, bool psf, float** dev) {
    std::vector<float> tb(TB_TOTAL, 0.f);
    for (int i = 0; i < WIN; i++) tb[TB_WIN + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / (psf ? WIN - 1 : WIN)));   // np.hamming | periodic
    for (int m = 0; m < 256; m++) {
        tb[TB_TW256 + 2 * m] = (float)cos(-2.0 * M_PI * m / 256.0);
        tb[TB_TW256 + 2 * m + 1] = (float)sin(-2.0 * M_PI * m / 256.0);
        tb[TB_TW512 + 2 * m] = (float)cos(-2.0 * M_PI * m / 512.0);
        tb[TB_TW512 + 2 * m + 1] = (float)sin(-2.0 * M_PI * m / 512.0);
    }
    if (psf) for (int i = 0; i < WIN; i++) tb[TB_WIN512 + i] = tb[TB_WIN + i];                        // frame at the start of the buffer, int16 units
    else for (int i = 0; i < WIN; i++) tb[TB_WIN512 + WOFF + i] = tb[TB_WIN + i] * (1.f / 32768.f);   // centred; exact: power of two
    const double sr = 16000.0;
    const double m_min = 0.0, m_max = 2595.0 * log10(1.0 + (sr / 2.0) / 700.0);
    double fpts[NMEL + 2];
    for (int i = 0; i < NMEL + 2; i++) {
        const double m = m_min + (m_max - m_min) * i / (NMEL + 1);
        fpts[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0);
    }
    std::vector<double> fb((size_t)NFREQ * NMEL, 0.0);
    for (int m = 0; m < NMEL; m++)
        for (int k = 0; k < NFREQ; k++) {
            double w;
            if (psf) {
                // get_filterbanks(40, 512, 16000, 0, 8000): triangles between the bins floor((nfft + 1) hz / samplerate),
                // times the 1 / nfft of powspec
                const double b0 = floor((NFFT + 1) * fpts[m] / sr), b1 = floor((NFFT + 1) * fpts[m + 1] / sr),
                             b2 = floor((NFFT + 1) * fpts[m + 2] / sr);
                w = 0.0;
                if (k >= b0 && k < b1) w = (k - b0) / (b1 - b0);
                else if (k >= b1 && k < b2) w = (b2 - k) / (b2 - b1);
                w *= 1.0 / NFFT;
            } else {
                // HTK mel triangles: torchaudio.functional.melscale_fbanks(257, 0, 8000, 40, 16000, None, 'htk')
                const double f = (sr / 2.0) * k / (NFREQ - 1);
                const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
                const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
                w = fmax(0.0, fmin(down, up));
            }
            fb[(size_t)k * NMEL + m] = w;
            // every non-zero weight must fall inside the band its filter tile reads
            const int tl = m / 16;
            if (w > 0.0 && (k < kMelK0[tl] || k >= kMelK0[tl] + 4 * kMelSteps[tl])) {
                VAR_SET_ERR(c, "mfcc tables: filter %d bin %d outside its band", m, k);
                return VAR_ERR_ARG;
            }
        }
    // B operands of v_mfma_f32_16x16x4_f32 (B[k = lane >> 4][col = lane & 15]), per wave and slot
    for (int wave = 0; wave < 4; wave++) {
        int slot = 0;
        for (int tl = 0; tl < 3; tl++)
            for (int q = 0; q < kMelSteps[tl] / 4; q++, slot++)
                for (int lane = 0; lane < 64; lane++) {
                    const int k = kMelK0[tl] + 4 * (4 * q + wave) + (lane >> 4), m = 16 * tl + (lane & 15);
                    tb[TB_MELB + (wave * kMelSlots + slot) * 64 + lane] = (k < NFREQ && m < NMEL) ? (float)(0.25 * fb[(size_t)k * NMEL + m]) : 0.f;   // the power tile holds 4 |X|^2
                }
    }
    for (int tl = 0; tl < 3; tl++)
        for (int s = 0; s < 10; s++)
            for (int lane = 0; lane < 64; lane++) {
                const int n = 4 * s + (lane >> 4), k = 16 * tl + (lane & 15);
                double v = 0.0;
                if (k < NMFCC) {
                    v = cos(M_PI / NMEL * (n + 0.5) * k) * sqrt(2.0 / NMEL);
                    if (k == 0) v *= 1.0 / sqrt(2.0);
                    if (psf) v *= 1.0 + (22.0 / 2.0) * sin(M_PI * k / 22.0);          // lifter(cepstra, L = 22)
                }
                tb[TB_DCTB + (tl * 10 + s) * 64 + lane] = (float)v;
            }
    static_assert(TB_TOTAL % 4 == 0, "table size");
    VAR_HIP_CHECK(c, hipMalloc((void**)dev, sizeof(float) * TB_TOTAL));
    VAR_HIP_CHECK(c, hipMemcpy(*dev, tb.data(), sizeof(float) * TB_TOTAL, hipMemcpyHostToDevice));
    if (psf) VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)mfcc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MFCC_LDS_BYTES));
    else VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)mfcc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MFCC_LDS_BYTES));
    return VAR_OK;
}

int mfcc_build_tables(var_ctx* c) { return build_tables(c, false, &c->mfcc_tab); }

template <bool PSF>
static int launch_flavour(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                          int pcm_stride, int out_frames, const float* tab, float* out, const char* who) {
    const long total = (long)nclips * out_frames;
    if (total > (1L << 23)) { VAR_SET_ERR(c, "%s: %d clips x %d frames is too many", who, nclips, out_frames); return VAR_ERR_ARG; }
    const int ntiles = (int)((total + FT - 1) / FT);
    const int grid = ntiles < 768 ? ntiles : 768;              // persistent: three 44 KB workgroups per CU
    hipLaunchKernelGGL(mfcc_kernel<PSF>, dim3(grid), dim3(NTHR), MFCC_LDS_BYTES, s, pcm, lens, clip_index, pcm_stride, out_frames,
                       (int)total, tab, out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// python_speech_features flavour (var_mfcc_psf, mfcc_psf.hip); its tables are built on first use
int launch_mfcc_psf(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                    int pcm_stride, int out_frames, float* out) {
    if (!c->mfcc_psf_tab) { const int r = build_tables(c, true, &c->mfcc_psf_tab); if (r != VAR_OK) return r; }
    return launch_flavour<true>(c, s, pcm, lens, clip_index, nclips, pcm_stride, out_frames, c->mfcc_psf_tab, out, "var_mfcc_psf");
}

int launch_mfcc(var_ctx* c, hipStream_t s, const int16_t* pcm, const int* lens, const int* clip_index, int nclips,
                int pcm_stride, int out_frames, float* out) {
    ProfScope prof(c, s, TAG_MFCC);
    return launch_flavour<false>(c, s, pcm, lens, clip_index, nclips, pcm_stride, out_frames, c->mfcc_tab, out, "var_mfcc");
}
