// Weight re-layout ("pack") and the fused flat-arena Adam step.
//   pack : OIHW parameters -> the [k][n] / [tap][n][c] images the conv kernels read
//   adam : torch.optim.Adam(...).step() (VAR/pretext_VAR.py:33-35,69) over one flat arena
#include "var_common.h"

struct PackSeg {
    int dst;        // offset in wpack
    int count;      // elements in this segment
    int src;        // offset of the source tensor in the parameter arena
    int cin, cout, taps;
    int kvalid;     // cin*taps (rows beyond it are zero padding)
    int type;       // 0: F image/1-D conv  [k=tap*cin+c][n] ; 1: D [tap][n][c] ; 2: F with source [n][tap][c]
};
struct PackTable { PackSeg seg[20]; int nseg; };

__global__ void __launch_bounds__(256) pack_weights_kernel(PackTable T, const float* __restrict__ params,
                                                            float* __restrict__ wpack, int total) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
        int si = 0;
#pragma unroll
        for (int i = 1; i < 20; ++i) if (i < T.nseg && j >= T.seg[i].dst) si = i;
        const PackSeg S = T.seg[si];
        const int e = j - S.dst;
        float v = 0.f;
        if (S.type == 1) {
            const int c = e % S.cin, n = (e / S.cin) % S.cout, tap = e / (S.cin * S.cout);
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

int launch_pack_weights(var_ctx* c, hipStream_t s, const float* params) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    PackTable T{};
    int n = 0;
    auto add = [&](int dst, int count, int src, int cin, int cout, int taps, int type) {
        T.seg[n++] = PackSeg{dst, count, src, cin, cout, taps, cin * taps, type};
    };
    for (int i = 0; i < 5; i++) {
        int Kp = kImgCh[i] * 9; if (Kp & 1) Kp++;
        add(K.img_f[i], Kp * kImgCh[i + 1], L.img_w[i], kImgCh[i], kImgCh[i + 1], 9, 0);
    }
    for (int i = 1; i < 5; i++) add(K.img_d[i], 9 * kImgCh[i + 1] * kImgCh[i], L.img_w[i], kImgCh[i], kImgCh[i + 1], 9, 1);
    add(K.snd_f[0], 200 * 32, L.snd_w[0], 40, 32, 5, 2);
    for (int i = 1; i < 4; i++) add(K.snd_f[i], 96 * 32, L.snd_w[i], 32, 32, 3, 0);
    for (int i = 1; i < 4; i++) add(K.snd_d[i], 96 * 32, L.snd_w[i], 32, 32, 3, 1);
    add(K.ih_w0t, kImgFeat * kHid, L.ih_w0, kImgFeat, kHid, 1, 0);   // plain transpose
    add(K.sh_w0t, kSndFeat * kHid, L.sh_w0, kSndFeat, kHid, 1, 0);
    T.nseg = n;   // 18
    ProfScope prof(c, s, TAG_PACK);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(256), dim3(256), 0, s, T, params, c->wpack, K.total);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    float b1, float b2, float eps, float wd,
                                                    float step_size, float bc2_sqrt) {
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
__global__ void adam_tick_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1; }

__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ lr_dev, float b1, float b2,
                                                        float eps, float wd, const int* __restrict__ step_dev) {
    const int t = step_dev[0];
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    const float step_size = (float)((double)lr_dev[0] / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

int launch_adam_dev(var_ctx* c, hipStream_t s, float* p, const float* g, float* m, float* v, long n,
                    const float* lr_dev, float b1, float b2, float eps, float wd, int* step_dev) {
    ProfScope prof(c, s, TAG_ADAM);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, s, step_dev);
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, lr_dev, b1, b2, eps, wd, step_dev);
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
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, b1, b2, eps, wd, step_size, bc2_sqrt);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
