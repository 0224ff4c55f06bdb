// Weight re-layout ("pack") and the fused flat-arena Adam step.
//   pack : OIHW parameters -> the [k][n] / [tap][n][c] images the conv kernels read
//   adam : torch.optim.Adam(...).step() (VAR/pretext_VAR.py:33-35,69) over one flat arena
#include <stdlib.h>

#include "var_common.h"

// ---- zero / copy by kernel (var_common.h: why not hipMemsetAsync / hipMemcpyAsync) ----
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
__global__ void __launch_bounds__(256) copy_words_kernel(unsigned* __restrict__ d, const unsigned* __restrict__ s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
static inline int words_grid(size_t n) { const size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }
int var_zero_async(var_ctx* c, hipStream_t s, void* p, size_t bytes) {
    if (!bytes) return VAR_OK;
    if ((bytes & 3) || ((uintptr_t)p & 3)) { VAR_SET_ERR(c, "var_zero_async: %zu bytes at %p are not whole aligned words", bytes, p); return VAR_ERR_ARG; }
    hipLaunchKernelGGL(zero_words_kernel, dim3(words_grid(bytes / 4)), dim3(256), 0, s, (unsigned*)p, bytes / 4);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
int var_copy_async(var_ctx* c, hipStream_t s, void* dst, const void* src, size_t bytes) {
    if (!bytes) return VAR_OK;
    if ((bytes & 3) || ((uintptr_t)dst & 3) || ((uintptr_t)src & 3)) { VAR_SET_ERR(c, "var_copy_async: not whole aligned words"); return VAR_ERR_ARG; }
    hipLaunchKernelGGL(copy_words_kernel, dim3(words_grid(bytes / 4)), dim3(256), 0, s, (unsigned*)dst, (const unsigned*)src, bytes / 4);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

struct PackSeg {
    int dst;        // offset in wpack
    int count;      // elements in this segment
    int src;        // offset of the source tensor in the parameter arena
    int cin, cout, taps;
    int kvalid;     // cin*taps (rows beyond it are zero padding)
    int type;       // 0: F image/1-D conv  [k=tap*cin+c][n] ; 1: D [tap][n][c] ; 2: F with source [n][tap][c] ;
                    // 3: D in MFMA A-fragment order [tap][c/16][n/16][16 (n&3) + (c&15)][(n>>2)&3]  (var_common.h: img_a)
                    // 4, 5: F in MFMA A-fragment order for D[n][pixel] (img_mid3.hip): [G][n/16][lane = 16 q + (n&15)][j] holds
                    //       W[n][c = 16 cg + 4 j + q][tap] -- one 1-KiB piece per (group of 16 k, n tile), a lane's 16 bytes = its
                    //       A operands of four k-steps; G = tap * (cin/16) + cg (type 4) or cg * 9 + tap (type 5: channel groups outermost)
};
struct PackTable { PackSeg seg[24]; int nseg; };

__global__ void __launch_bounds__(256) pack_weights_kernel(PackTable T, const float* __restrict__ params,
                                                            float* __restrict__ wpack, int total) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
        int si = 0;
#pragma unroll
        for (int i = 1; i < 24; ++i) if (i < T.nseg && j >= T.seg[i].dst) si = i;
        const PackSeg S = T.seg[si];
        const int e = j - S.dst;
        float v = 0.f;
        if (S.type == 1) {
            const int c = e % S.cin, n = (e / S.cin) % S.cout, tap = e / (S.cin * S.cout);
            v = params[S.src + (n * S.cin + c) * S.taps + tap];
        } else if (S.type == 3) {
            const int jj = e & 3, ln = (e >> 2) & 63, grp = e >> 8;
            const int nsg = S.cout / 16, nct = S.cin / 16;
            const int sg = grp % nsg, ct = (grp / nsg) % nct, tap = grp / (nsg * nct);
            const int n = 16 * sg + 4 * jj + (ln >> 4), c = 16 * ct + (ln & 15);
            v = params[S.src + (n * S.cin + c) * S.taps + tap];
        } else if (S.type >= 4) {
            const int jj = e & 3, ln = (e >> 2) & 63, grp = e >> 8;
            const int nnt = S.cout / 16, ncg = S.cin / 16;
            const int nt = grp % nnt, G = grp / nnt;
            const int cg = S.type == 4 ? G % ncg : G / 9, tap = S.type == 4 ? G / ncg : G % 9;
            const int n = 16 * nt + (ln & 15), c = 16 * cg + 4 * jj + (ln >> 4);
            v = params[S.src + (n * S.cin + c) * S.taps + tap];
        } else {
            const int n = e % S.cout, k = e / S.cout;
            if (k < S.kvalid) {
                const int tap = k / S.cin, c = k - tap * S.cin;
                v = (S.type == 0) ? params[S.src + (n * S.cin + c) * S.taps + tap]
                                  : params[S.src + (n * S.taps + tap) * S.cin + c];
            }
        }
        wpack[j] = v;
    }
}

static PackTable make_pack_table(var_ctx* c) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    PackTable T{};
    int n = 0;
    auto add = [&](int dst, int count, int src, int cin, int cout, int taps, int type) {
        T.seg[n++] = PackSeg{dst, count, src, cin, cout, taps, cin * taps, type};
    };
    for (int i = 0; i < 5; i++) {
        int Kp = kImgCh[i] * 9; if (Kp & 1) Kp++;
        // conv 1, 2: [k][n]; conv 3..5: A-fragment pieces for img_mid3.hip (conv 3 with its channel groups outermost)
        add(K.img_f[i], Kp * kImgCh[i + 1], L.img_w[i], kImgCh[i], kImgCh[i + 1], 9, i < 2 ? 0 : (i == 2 ? 5 : 4));
    }
    add(K.img_d[1], 9 * kImgCh[2] * kImgCh[1], L.img_w[1], kImgCh[1], kImgCh[2], 9, 1);    // (layers 2..4: img_a, the chain's form)
    // (segments must stay sorted by dst for the gather kernel's search: img_a sits between the sound images and the heads in PackLayout)
    add(K.snd_f[0], 200 * 32, L.snd_w[0], 40, 32, 5, 2);
    for (int i = 1; i < 4; i++) add(K.snd_f[i], 96 * 32, L.snd_w[i], 32, 32, 3, 0);
    for (int i = 1; i < 4; i++) add(K.snd_d[i], 96 * 32, L.snd_w[i], 32, 32, 3, 1);
    for (int i = 2; i < 5; i++) add(K.img_a[i], 9 * kImgCh[i + 1] * kImgCh[i], L.img_w[i], kImgCh[i], kImgCh[i + 1], 9, 3);
    add(K.ih_w0t, kImgFeat * kHid, L.ih_w0, kImgFeat, kHid, 1, 0);   // plain transpose
    add(K.sh_w0t, kSndFeat * kHid, L.sh_w0, kSndFeat, kHid, 1, 0);
    T.nseg = n;   // 21
    return T;
}

int launch_pack_weights(var_ctx* c, hipStream_t s, const float* params) {
    const PackLayout& K = c->kl;
    const PackTable T = make_pack_table(c);
    ProfScope prof(c, s, TAG_PACK);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(256), dim3(256), 0, s, T, params, c->wpack, K.total);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    float b1, float b2, float eps, float wd,
                                                    float step_size, float bc2_sqrt, const unsigned* __restrict__ guard,
                                                    const float* __restrict__ gloss) {
    if (guard && *guard) return;          // this step's gradient is invalid (var_ctx::adam_guard): leave the state alone
    if (gloss && !(fabsf(*gloss) <= 3.4e38f)) return;      // ... or some rank's was: its NaN came in with the all-reduced loss
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);           // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;          // mul_(beta2).addcmul_(g, g, 1-beta2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// Capturable variant (HIP graphs): the step count and the learning rate live in device memory, the
// bias corrections are formed on the device, so a captured launch stays correct on every replay.
// ---- the graph-replayed step: Adam + weight re-pack + step counter + next index row, ONE launch ----
// Every parameter element is updated once and SCATTERED to the (up to two) packed images that hold it -- the
// inverse of pack_weights_kernel's gather, so the separate re-pack launch and its read of the arena go away.
// The step count is read by every block at its start and advanced by the last block to finish (a device
// counter tells which one that is), so the launch can be captured into a HIP graph without a "tick" kernel.
// Block 0 also copies the next row of the epoch's index table into the row buffer the step's kernels read,
// so a replay needs no host-side copy at all.
constexpr int kAdamT = 1024;      // threads per block: one element per thread at this model's size, one round of loads
__global__ void __launch_bounds__(kAdamT)
adam_pack_dev_kernel(const PackSeg* __restrict__ segs, int nseg, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                     float* __restrict__ v, long n, const float* __restrict__ lr_dev, float b1, float b2, float eps,
                     float wd, int* __restrict__ step_dev, unsigned* __restrict__ done_ctr, float* __restrict__ wpack,
                     const int* __restrict__ idx_table, int row_ints, int n_rows, int* __restrict__ cursor,
                     int* __restrict__ idx_row, int ahead_from, const unsigned* __restrict__ guard, const float* __restrict__ gloss) {
    if (guard && *guard) return;          // (every block: no update, no step count, no cursor move)
    if (gloss && !(fabsf(*gloss) <= 3.4e38f)) return;
    // bias corrections: double-precision pow once per block, not per thread
    __shared__ float sh_step[2];
    const int t = step_dev[0] + 1;
    if (threadIdx.x == 0) {
        // beta^t by repeated squaring in double (t is an integer): ~40 multiplies instead of a software pow();
        // agrees with pow() to a few ulp of double, far below the float the result is rounded to
        auto ipow = [](double b, int e) { double r = 1.0; while (e > 0) { if (e & 1) r *= b; b *= b; e >>= 1; } return r; };
        const double bc1 = 1.0 - ipow((double)b1, t);
        const double bc2 = 1.0 - ipow((double)b2, t);
        sh_step[0] = (float)((double)lr_dev[0] / bc1);
        sh_step[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = sh_step[0], bc2_sqrt = sh_step[1];
    if (blockIdx.x == 0 && idx_table) {
        int next = cursor[0] + 1;
        if (next >= n_rows) next = 0;
        int next2 = next + 1;                                // ahead: a second copy, one row further on, behind the first
        if (next2 >= n_rows) next2 = 0;
        for (int e = threadIdx.x; e < row_ints; e += kAdamT) {
            idx_row[e] = idx_table[(size_t)next * row_ints + e];
            if (ahead_from > 0) idx_row[row_ints + e] = idx_table[(size_t)next2 * row_ints + e];
        }
        __syncthreads();
        if (threadIdx.x == 0) cursor[0] = next;
    }
    // segment table -> LDS (one copy per block); each wave then finds the segments that touch its 64 consecutive
    // elements with ONE ballot (lane q tests segment q) instead of scanning the table per element
    __shared__ PackSeg ssegs[24];
    if (wpack) {
        for (int e = threadIdx.x; e < nseg * (int)(sizeof(PackSeg) / 4); e += kAdamT) ((int*)ssegs)[e] = ((const int*)segs)[e];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    for (long w0 = (long)blockIdx.x * kAdamT + (threadIdx.x - lane); w0 < n; w0 += (long)gridDim.x * kAdamT) {   // wave-uniform
        const long i = w0 + lane;
        float pn = 0.f;
        if (i < n) {
            const float pi = p[i];
            const float gi = g[i] + wd * pi;
            const float mi = m[i] + (gi - m[i]) * (1.f - b1);
            const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pn = pi - step_size * (mi / denom);
            p[i] = pn;
        }
        if (wpack) {
            bool hit = false;
            if (lane < nseg) {
                const int src = ssegs[lane].src, cnt = ssegs[lane].cin * ssegs[lane].cout * ssegs[lane].taps;
                hit = src < w0 + 64 && src + cnt > w0;
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int q = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const PackSeg S = ssegs[q];
                const int e = (int)i - S.src;
                if (i >= n || e < 0 || e >= S.cin * S.cout * S.taps) continue;
                // (e, divisors < 2^18: quotient through a float reciprocal, then one exact correction step)
                auto divmod = [](int a, int d, int& qq, int& r) {
                    qq = (int)((float)a * __frcp_rn((float)d));
                    r = a - qq * d;
                    if (r < 0) { --qq; r += d; } else if (r >= d) { ++qq; r -= d; }
                };
                int c, tap, nn, rest;
                if (S.type == 2) { divmod(e, S.cin, rest, c); divmod(rest, S.taps, nn, tap); }
                else { divmod(e, S.taps, rest, tap); divmod(rest, S.cin, nn, c); }
                int d;
                if (S.type == 3)
                    d = ((((tap * (S.cin / 16) + (c >> 4)) * (S.cout / 16) + (nn >> 4)) * 64 + 16 * (nn & 3) + (c & 15)) << 2) + ((nn >> 2) & 3);
                else if (S.type >= 4) {
                    const int G = S.type == 4 ? tap * (S.cin / 16) + (c >> 4) : (c >> 4) * 9 + tap;
                    d = (((G * (S.cout / 16) + (nn >> 4)) * 64 + 16 * (c & 3) + (nn & 15)) << 2) + ((c >> 2) & 3);
                } else d = S.type == 1 ? (tap * S.cout + nn) * S.cin + c : (tap * S.cin + c) * S.cout + nn;
                wpack[S.dst + d] = pn;
            }
        }
    }
    // The last block to get here has seen every block read step_dev[0]: the increment carries a data dependency
    // on the value read (t is never INT_MIN), so no fence is needed -- a device-scope fence here would write back
    // the whole L2 once per block.
    if (threadIdx.x == 0) {
        if (atomicAdd(done_ctr, (unsigned)(t != (int)0x80000000)) == gridDim.x - 1) { step_dev[0] = t; done_ctr[0] = 0; }
    }
}

// the segment table, once (var_init), in device memory
int pack_table_upload(var_ctx* c) {
    const PackTable T = make_pack_table(c);
    VAR_HIP_CHECK(c, hipMalloc((void**)&c->pack_segs_dev, sizeof(T.seg)));
    VAR_HIP_CHECK(c, hipMemcpy(c->pack_segs_dev, T.seg, sizeof(T.seg), hipMemcpyHostToDevice));
    c->pack_nseg = T.nseg;
    return VAR_OK;
}

int launch_adam_dev(var_ctx* c, hipStream_t s, float* p, const float* g, float* m, float* v, long n,
                    const float* lr_dev, float b1, float b2, float eps, float wd, int* step_dev, bool repack,
                    const int* idx_table, int row_ints, int n_rows, int* cursor, int* idx_row, int ahead_from) {
    ProfScope prof(c, s, TAG_ADAM);
    // few blocks: every block ends with one device-scope atomic on the same word
    int grid = (int)((n + kAdamT - 1) / kAdamT);
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(adam_pack_dev_kernel, dim3(grid), dim3(kAdamT), 0, s, (const PackSeg*)c->pack_segs_dev, c->pack_nseg, p, g, m, v, n, lr_dev,
                       b1, b2, eps, wd, step_dev, c->done_ctr, repack ? c->wpack : nullptr, idx_table, row_ints, n_rows,
                       cursor, idx_row, ahead_from, (c->adam_guard && n == c->adam_guard_n) ? c->adam_guard : nullptr,
                       n == c->adam_guard_n ? c->adam_guard_loss : nullptr);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int launch_adam(var_ctx* c, hipStream_t s, float* p, const float* g, float* m, float* v, long n,
                float lr, float b1, float b2, float eps, float wd, int step) {
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    ProfScope prof(c, s, TAG_ADAM);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, b1, b2, eps, wd, step_size, bc2_sqrt,
                       (c->adam_guard && n == c->adam_guard_n) ? c->adam_guard : nullptr, n == c->adam_guard_n ? c->adam_guard_loss : nullptr);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
