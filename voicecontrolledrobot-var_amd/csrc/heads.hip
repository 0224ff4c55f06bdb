// Embedding heads, L2 normalisation and the triplet loss, forward and backward:
//   imgTriplet / soundTriplet = Linear(K,128)+ReLU+Linear(128,3)   (arm_pretext_model.py:46-56)
//   F.normalize(p=2, dim=1, eps=1e-12)                              (pretext_base.py:18,23)
//   TripletMarginLoss(margin, p=2, eps=1e-6, mean)                  (VAR/pretext_VAR.py:38,64)
// Rows are samples; a workgroup owns 8 rows, keeps them in LDS and walks the transposed
// first-layer weight W0T[k][j] (packed image) with coalesced loads, each value reused for
// its 4 rows per thread.  All reductions are fixed-order (bitwise reproducible).
#include "var_common.h"

namespace {
constexpr int RB = 8;   // rows per workgroup

template <int K>
__global__ void __launch_bounds__(256)
heads_fwd_kernel(const float* __restrict__ x, int nrows, int row_lo, int row_hi,
                 const float* __restrict__ w0t, const float* __restrict__ b0,
                 const float* __restrict__ w1, const float* __restrict__ b1,
                 float* __restrict__ hid, float* __restrict__ emb_raw, float* __restrict__ emb) {
    __shared__ float xs[RB * K];
    __shared__ float hs[RB * kHid];
    __shared__ float raw[RB * 4];
    const int tid = threadIdx.x, j = tid & 127, rh = tid >> 7;
    const int r0 = blockIdx.x * RB;
    for (int e = tid; e < RB * K; e += 256) {
        const int r = e / K, k = e - r * K, row = r0 + r;
        xs[e] = (row >= row_lo && row < row_hi) ? x[(size_t)row * K + k] : 0.f;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        const float wv = w0t[k * kHid + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += wv * xs[(rh * 4 + i) * K + k];
    }
    const float bv = b0[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rh * 4 + i, row = r0 + r;
        float v = acc[i] + bv;
        v = v > 0.f ? v : 0.f;
        hs[r * kHid + j] = v;
        if (row >= row_lo && row < row_hi) hid[(size_t)row * kHid + j] = v;
    }
    __syncthreads();
    if (tid < RB * 3) {
        const int r = tid / 3, d = tid - r * 3;
        float s = b1[d];
        for (int jj = 0; jj < kHid; ++jj) s += hs[r * kHid + jj] * w1[d * kHid + jj];
        raw[r * 4 + d] = s;
    }
    __syncthreads();
    if (tid < RB) {
        const int row = r0 + tid;
        if (row >= row_lo && row < row_hi) {
            const float a = raw[tid * 4], b = raw[tid * 4 + 1], c = raw[tid * 4 + 2];
            const float nrm = sqrtf(a * a + b * b + c * c);
            const float den = nrm > 1e-12f ? nrm : 1e-12f;
            emb_raw[row * 3 + 0] = a; emb_raw[row * 3 + 1] = b; emb_raw[row * 3 + 2] = c;
            emb[row * 3 + 0] = a / den; emb[row * 3 + 1] = b / den; emb[row * 3 + 2] = c / den;
        }
    }
}

// loss_out[0] = inv_count * sum_i max(||a-p+eps|| - ||a-n+eps|| + margin, 0); grads of loss_out.
__global__ void __launch_bounds__(256)
triplet_kernel(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n, int B,
               float margin, float inv_count, float* __restrict__ loss_out,
               float* __restrict__ ga, float* __restrict__ gp, float* __restrict__ gn) {
    __shared__ float part[4];
    const int tid = threadIdx.x;
    float local = 0.f;
    for (int i = tid; i < B; i += 256) {
        float dp[3], dn[3], sp = 0.f, sn = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            dp[d] = (a[i * 3 + d] - p[i * 3 + d]) + 1e-6f;
            dn[d] = (a[i * 3 + d] - n[i * 3 + d]) + 1e-6f;
            sp += dp[d] * dp[d];
            sn += dn[d] * dn[d];
        }
        const float dap = sqrtf(sp), dan = sqrtf(sn);
        const float l = dap - dan + margin;
        const bool active = l > 0.f;
        if (active) local += l;
        const float s = active ? inv_count : 0.f;
        const float ip = dap > 0.f ? 1.f / dap : 0.f, in_ = dan > 0.f ? 1.f / dan : 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float up = dp[d] * ip, un = dn[d] * in_;
            if (ga) ga[i * 3 + d] = s * (up - un);
            if (gp) gp[i * 3 + d] = -s * up;
            if (gn) gn[i * 3 + d] = s * un;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((tid & 63) == 0) part[tid >> 6] = local;
    __syncthreads();
    if (tid == 0) loss_out[0] = ((part[0] + part[1]) + (part[2] + part[3])) * inv_count;
}

// Backward, per-row part: normalise-bwd -> graw; ghid = (graw @ W1) * (h>0); gx = (ghid @ W0) * (x>0)
template <int K>
__global__ void __launch_bounds__(256)
heads_bwd_rows_kernel(const float* __restrict__ x, int nrows, int row_lo, int row_hi,
                      const float* __restrict__ w0 /*(128,K)*/, const float* __restrict__ w1 /*(3,128)*/,
                      const float* __restrict__ hid, const float* __restrict__ emb_raw,
                      const float* __restrict__ emb, const float* __restrict__ gemb,
                      float* __restrict__ graw_out, float* __restrict__ ghid_out, float* __restrict__ gx) {
    __shared__ float gh[RB * kHid];
    __shared__ float gr[RB * 4];
    const int tid = threadIdx.x, j = tid & 127, rh = tid >> 7;
    const int r0 = blockIdx.x * RB;
    if (tid < RB) {
        const int row = r0 + tid;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (row >= row_lo && row < row_hi) {
            const float a = emb_raw[row * 3], b = emb_raw[row * 3 + 1], c = emb_raw[row * 3 + 2];
            const float nrm = sqrtf(a * a + b * b + c * c);
            const float den = nrm > 1e-12f ? nrm : 1e-12f;
            const float y0 = emb[row * 3], y1 = emb[row * 3 + 1], y2 = emb[row * 3 + 2];
            const float e0 = gemb[row * 3], e1 = gemb[row * 3 + 1], e2 = gemb[row * 3 + 2];
            const float dot = y0 * e0 + y1 * e1 + y2 * e2;
            g0 = (e0 - y0 * dot) / den; g1 = (e1 - y1 * dot) / den; g2 = (e2 - y2 * dot) / den;
            graw_out[row * 3] = g0; graw_out[row * 3 + 1] = g1; graw_out[row * 3 + 2] = g2;
        }
        gr[tid * 4] = g0; gr[tid * 4 + 1] = g1; gr[tid * 4 + 2] = g2;
    }
    __syncthreads();
    {
        const float wa = w1[j], wb = w1[kHid + j], wc = w1[2 * kHid + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rh * 4 + i, row = r0 + r;
            float v = 0.f;
            if (row >= row_lo && row < row_hi) {
                v = gr[r * 4] * wa + gr[r * 4 + 1] * wb + gr[r * 4 + 2] * wc;
                if (!(hid[(size_t)row * kHid + j] > 0.f)) v = 0.f;
                ghid_out[(size_t)row * kHid + j] = v;
            }
            gh[r * kHid + j] = v;
        }
    }
    __syncthreads();
    constexpr int KC = (K + 255) / 256;
    float acc[KC][RB];
#pragma unroll
    for (int q = 0; q < KC; ++q)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[q][r] = 0.f;
#pragma unroll 2
    for (int jj = 0; jj < kHid; ++jj) {
#pragma unroll
        for (int q = 0; q < KC; ++q) {
            const int k = tid + 256 * q;
            const float wv = (k < K) ? w0[jj * K + k] : 0.f;
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[q][r] += wv * gh[r * kHid + jj];
        }
    }
#pragma unroll
    for (int q = 0; q < KC; ++q) {
        const int k = tid + 256 * q;
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = r0 + r;
            if (row >= row_lo && row < row_hi) {
                const size_t o = (size_t)row * K + k;
                gx[o] = x[o] > 0.f ? acc[q][r] : 0.f;
            }
        }
    }
}

// Backward, weight part: dW0 = ghid^T x, db0 = sum ghid, dW1 = graw^T hid, db1 = sum graw
// over rows [row_lo,row_hi).  Blocks 0..K/32-1 own 32 columns of dW0 each; the last block
// owns the small tensors.  Fixed row order => deterministic.
template <int K>
__global__ void __launch_bounds__(256)
heads_bwd_w_kernel(const float* __restrict__ x, int row_lo, int row_hi,
                   const float* __restrict__ hid, const float* __restrict__ graw,
                   const float* __restrict__ ghid,
                   float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw1,
                   float* __restrict__ db1) {
    __shared__ float ghs[16 * kHid];
    __shared__ float xs[16 * 32];
    const int tid = threadIdx.x;
    if (blockIdx.x == K / 32) {
        if (tid < kHid) {
            float s = 0.f;
            for (int r = row_lo; r < row_hi; ++r) s += ghid[(size_t)r * kHid + tid];
            db0[tid] = s;
        }
        for (int o = tid; o < 3 * kHid; o += 256) {
            const int d = o / kHid, jj = o - d * kHid;
            float s = 0.f;
            for (int r = row_lo; r < row_hi; ++r) s += graw[r * 3 + d] * hid[(size_t)r * kHid + jj];
            dw1[o] = s;
        }
        if (tid < 3) {
            float s = 0.f;
            for (int r = row_lo; r < row_hi; ++r) s += graw[r * 3 + tid];
            db1[tid] = s;
        }
        return;
    }
    const int k0 = blockIdx.x * 32, kk = tid & 31, jg = tid >> 5;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int rb = row_lo; rb < row_hi; rb += 16) {
        __syncthreads();
        for (int e = tid; e < 16 * kHid; e += 256) {
            const int r = rb + e / kHid;
            ghs[e] = r < row_hi ? ghid[(size_t)r * kHid + (e % kHid)] : 0.f;
        }
        for (int e = tid; e < 16 * 32; e += 256) {
            const int r = rb + e / 32;
            xs[e] = r < row_hi ? x[(size_t)r * K + k0 + (e % 32)] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            const float xv = xs[rr * 32 + kk];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += ghs[rr * kHid + jg * 16 + i] * xv;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) dw0[(size_t)(jg * 16 + i) * K + k0 + kk] = acc[i];
}
}  // namespace

int launch_heads_fwd(var_ctx* c, hipStream_t s, const float* params, int B, bool has_img, bool has_pos, bool has_neg) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    ProfScope prof(c, s, TAG_HEADS_FWD);
    if (has_img) {
        hipLaunchKernelGGL(heads_fwd_kernel<kImgFeat>, dim3((B + RB - 1) / RB), dim3(256), 0, s,
                           c->act[5], B, 0, B, c->wpack + K.ih_w0t, params + L.ih_b0, params + L.ih_w1,
                           params + L.ih_b1, c->hid_i, c->emb_raw, c->emb);
    }
    if (has_pos || has_neg) {
        const int lo = has_pos ? 0 : B, hi = has_neg ? 2 * B : B;
        hipLaunchKernelGGL(heads_fwd_kernel<kSndFeat>, dim3((2 * B + RB - 1) / RB), dim3(256), 0, s,
                           c->sact[4], 2 * B, lo, hi, c->wpack + K.sh_w0t, params + L.sh_b0, params + L.sh_w1,
                           params + L.sh_b1, c->hid_s, c->emb_raw + 3 * B, c->emb + 3 * B);
    }
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

int launch_triplet(var_ctx* c, hipStream_t s, const float* a, const float* p, const float* n, int B,
                   float margin, float inv_count, float* loss_out, float* ga, float* gp, float* gn) {
    ProfScope prof(c, s, TAG_TRIPLET);
    hipLaunchKernelGGL(triplet_kernel, dim3(1), dim3(256), 0, s, a, p, n, B, margin, inv_count, loss_out, ga, gp, gn);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// gemb (3B,3) must hold the gradients wrt the normalised embeddings [img | pos | neg].
// Produces gact[5] (B,576), gsact[4] (2B,160) and the 8 head gradient tensors.
int launch_heads_bwd(var_ctx* c, hipStream_t s, const float* params, float* grads, int B, bool has_img,
                     int snd_lo, int snd_hi) {
    const ParamLayout& L = c->pl;
    float* graw = c->gemb + 9 * (size_t)c->maxB;          // second half of the gemb buffer
    ProfScope prof(c, s, TAG_HEADS_BWD_ROWS);
    if (has_img) {
        hipLaunchKernelGGL(heads_bwd_rows_kernel<kImgFeat>, dim3((B + RB - 1) / RB), dim3(256), 0, s,
                           c->act[5], B, 0, B, params + L.ih_w0, params + L.ih_w1, c->hid_i, c->emb_raw, c->emb,
                           c->gemb, graw, c->ghid, c->gact[5]);
        hipLaunchKernelGGL(heads_bwd_w_kernel<kImgFeat>, dim3(kImgFeat / 32 + 1), dim3(256), 0, s,
                           c->act[5], 0, B, c->hid_i, graw, c->ghid,
                           grads + L.ih_w0, grads + L.ih_b0, grads + L.ih_w1, grads + L.ih_b1);
    }
    if (snd_hi > snd_lo) {
        hipLaunchKernelGGL(heads_bwd_rows_kernel<kSndFeat>, dim3((2 * B + RB - 1) / RB), dim3(256), 0, s,
                           c->sact[4], 2 * B, snd_lo, snd_hi, params + L.sh_w0, params + L.sh_w1, c->hid_s,
                           c->emb_raw + 3 * B, c->emb + 3 * B, c->gemb + 3 * B, graw + 3 * B,
                           c->ghid + (size_t)B * kHid, c->gsact[4]);
        hipLaunchKernelGGL(heads_bwd_w_kernel<kSndFeat>, dim3(kSndFeat / 32 + 1), dim3(256), 0, s,
                           c->sact[4], snd_lo, snd_hi, c->hid_s, graw + 3 * B, c->ghid + (size_t)B * kHid,
                           grads + L.sh_w0, grads + L.sh_b0, grads + L.sh_w1, grads + L.sh_b1);
    }
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
