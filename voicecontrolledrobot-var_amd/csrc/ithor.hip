// iTHOR VARPretextNet on gfx950 (models/pretext/ai2thor_pretext_model.py:5-58):
//   image   (B,3,H,H)  -> conv3x3 s1 x2, pool, conv, pool, conv, pool, conv, pool, conv3x3 s2 -> (B,1152)
//   sound   (B,1,600,40) x {pos,neg} -> conv 11x11 s2, conv 11x5 s2, conv 7x3 s2 -> (B,73,448) -> bidirectional
//           GRU(448 -> 512), final hidden states of both directions concatenated -> (B,1024)
//   heads   Linear(1152,128)+ReLU+Linear(128,3); Linear(1024,128)+ReLU+Linear(128,64)+ReLU+Linear(64,3); L2-normalise
// forward, TripletMarginLoss and the full backward.  Every product (convolutions in all three directions, Linear
// layers, the GRU's input and recurrent products and their gradients) is an instance of the f32-MFMA
// gather-GEMM of gg.h; this file adds the element-wise kernels (max pool, GRU gates, ReLU masks, bias sums,
// normalise) and the host-side schedule.  Parameters are used in place in their state_dict() layouts.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gg.h"
#include "dense_bf16.h"

// ---- geometry ------------------------------------------------------------------------------------------------
static constexpr int kICh[7] = {3, 32, 32, 64, 64, 128, 128};
static constexpr int kT = 600, kF = 40;             // sound_dim (1,600,40), Envs/ai2thor/config.py
static constexpr int kSeq = 73, kGin = 448, kGh = 512, kG3 = 1536;
static constexpr int kIRaw = 1152, kSRaw = 1024;

struct IthorLayout {
    int iw[6], ib[6];
    int w_ih[2], w_hh[2], b_ih[2], b_hh[2];
    int sw[3], sb[3];
    int ih_w0, ih_b0, ih_w1, ih_b1;
    int sh_w0, sh_b0, sh_w1, sh_b1, sh_w2, sh_b2;
    int total;
};
static constexpr int kSK[3] = {121, 64 * 55, 64 * 21};

static IthorLayout make_ithor_layout() {
    IthorLayout L{};
    int o = 0;
    for (int i = 0; i < 6; i++) { L.iw[i] = o; o += kICh[i + 1] * kICh[i] * 9; L.ib[i] = o; o += kICh[i + 1]; }
    for (int d = 0; d < 2; d++) {
        L.w_ih[d] = o; o += kG3 * kGin; L.w_hh[d] = o; o += kG3 * kGh;
        L.b_ih[d] = o; o += kG3;        L.b_hh[d] = o; o += kG3;
    }
    for (int i = 0; i < 3; i++) { L.sw[i] = o; o += 64 * kSK[i]; L.sb[i] = o; o += 64; }
    L.ih_w0 = o; o += 128 * kIRaw; L.ih_b0 = o; o += 128; L.ih_w1 = o; o += 3 * 128; L.ih_b1 = o; o += 3;
    L.sh_w0 = o; o += 128 * kSRaw; L.sh_b0 = o; o += 128; L.sh_w1 = o; o += 64 * 128; L.sh_b1 = o; o += 64;
    L.sh_w2 = o; o += 3 * 64;      L.sh_b2 = o; o += 3;
    L.total = o;
    return L;
}

struct ithor_state {
    IthorLayout L;
    int maxB = 0, H = 0;
    int hs[6] = {0};                // image side lengths: input, after pool 1..4, after the last conv
    char* ws = nullptr;
    // image activations (post-ReLU) and pooled maps, and the gradients wrt them
    float *a[7] = {nullptr}, *p[6] = {nullptr}, *ga[7] = {nullptr}, *gp[6] = {nullptr};
    float *s[4] = {nullptr}, *gs[4] = {nullptr};          // sound conv outputs 1..3 (3 in sequence layout)
    float *GI = nullptr, *GH = nullptr, *Hb = nullptr, *R = nullptr, *Z = nullptr, *Nn = nullptr, *GHN = nullptr;
    float *DGI = nullptr, *DGH = nullptr, *DH = nullptr, *DHP = nullptr;
    float *slab = nullptr, *bslab = nullptr;              // split-K partial sums / bias-sum partials
    long bs_used = 0;                                     // bslab is handed out in pieces (bs_take) so that the folds can wait
    void* folds = nullptr;                                // FoldJobs*: folds of bslab partials, run together (flush_folds)
    // bf16 mode, layer 6 as a dense layer (l6_*): the expanded filter matrix, its bias, bf16 copies of the two activations
    unsigned short *l6w = nullptr, *l6x = nullptr, *l6g = nullptr;
    float* l6b = nullptr;
    bool img_packed = false;                              // this forward packed the filters of layers 2-5 (img_bf16_pack_all)
    bool l6_ready = false;                                // this forward built l6w / l6b (the backward re-uses them)
    void* imgws = nullptr;                                // fragment-ordered filters of img_bf16.hip
    void* gruws = nullptr;                                // W_hh in MFMA fragment order (gru_bf16.hip)
    void* bfws = nullptr;                                 // bf16 images / packed filters of the staged sound kernels (snd_bf16.hip)
    int gh_split = 1, dh_split = 1;
    bool gru_drop_one = false;                            // tests: the next persistent forward misses a workgroup per group
    bool gru_seq = true;                                  // bf16 mode: each GRU pass as one persistent launch (gru_bf16.hip)
    bool bf16 = false, keep32 = false;                    // bf16 mode; ... with the fp32 copies of the sound maps (tests)
    float *sraw = nullptr, *gsraw = nullptr;              // (clips,1024)
    float *hid_i = nullptr, *ghid_i = nullptr, *hid_s1 = nullptr, *ghid_s1 = nullptr, *hid_s2 = nullptr, *ghid_s2 = nullptr;
    float *raw = nullptr, *graw = nullptr, *emb = nullptr, *gemb = nullptr;   // (3B,3) [img | pos | neg]
    float* loss = nullptr;
    // saved forward
    int gen = 0;                                          // generation id of the saved forward (var_ithor_saved_generation)
    int B = 0, nclips = 0; bool has_img = false, has_pos = false, has_neg = false;
    const void* image = nullptr; int is_u8 = 0; long bstride = 0;
    const float *pos = nullptr, *neg = nullptr;
};

static inline ithor_state* ith(var_ctx* c) { return (ithor_state*)c->ith; }

// every product of the model goes through here: fp32 operands, or bf16 operands (fp32 accumulate) when the context
// was switched with var_ithor_set_bf16
template <int KC = GG_KC, class P>
static int gg(var_ctx* c, hipStream_t s, const P& p, int batches = 1);

static void ithor_update_guard(var_ctx* c);

void ithor_free(var_ctx* c) {
    ithor_state* st = ith(c);
    if (!st) return;
    if (st->ws) (void)hipFree(st->ws);
    free(st->folds);
    delete st;
    c->ith = nullptr;
    c->adam_guard = nullptr; c->adam_guard_n = 0; c->adam_guard_loss = nullptr;
}

template <int KC, class P>
static int gg(var_ctx* c, hipStream_t s, const P& p, int batches) {
    return ith(c)->bf16 ? gg_launch<P, KC, true>(c, s, p, batches) : gg_launch<P, KC, false>(c, s, p, batches);
}
// bf16 mode: operand copies (written by the GRU step kernels / gru_bf16_convert_x) instead of the fp32 arrays, where the
// staged kernel takes the resulting shape -- the big products are bound by their operand traffic
template <class P>
static void use_bf16_copies(var_ctx* c, P& p, const void* a16, const void* b16) {
    if (!ith(c)->bf16) return;
    P q = p;
    if (a16) { q.A = (const float*)a16; q.a16 = 1; }
    if (b16) { q.Bm = (const float*)b16; q.b16 = 1; }
    if (dense16::eligible(q)) p = q;
}
// dense products: in bf16 mode the LDS-staged kernel of dense_bf16.h where its alignment conditions hold
template <bool AK, bool BK, int MODE>
static int gg(var_ctx* c, hipStream_t s, const DenseP<AK, BK, MODE>& p, int batches = 1) {
    if (ith(c)->bf16 && dense16::eligible(p)) return dense16::launch(c, s, p, batches);
    return ith(c)->bf16 ? gg_launch<DenseP<AK, BK, MODE>, GG_KC, true>(c, s, p, batches)
                        : gg_launch<DenseP<AK, BK, MODE>, GG_KC, false>(c, s, p, batches);
}

// ---- element-wise kernels ---------------------------------------------------------------------------------------
// nn.MaxPool2d(2, stride=2): the last row/column of an odd map is dropped
__global__ void pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int H, int HP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = (int)(i % HP), py = (int)((i / HP) % HP);
    const long plane = i / ((long)HP * HP);
    const float* q = x + plane * H * H + (long)(2 * py) * H + 2 * px;
    y[i] = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[H], q[H + 1]));
}

// backward of ReLU followed by the max pool: the gradient of a pooled cell goes to the first maximum of its
// window in scan order (PyTorch's choice) and is dropped where that activation is not positive.
__global__ void pool_relu_bwd_kernel(const float* __restrict__ act, const float* __restrict__ gpool,
                                     float* __restrict__ gact, long n, int H, int HP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % H), y = (int)((i / H) % H);
    const long plane = i / ((long)H * H);
    const int px = x >> 1, py = y >> 1;
    float g = 0.f;
    if (px < HP && py < HP) {
        const float* q = act + plane * H * H + (long)(2 * py) * H + 2 * px;
        const float v0 = q[0], v1 = q[1], v2 = q[H], v3 = q[H + 1];
        int best = 0; float bv = v0;
        if (v1 > bv) { bv = v1; best = 1; }
        if (v2 > bv) { bv = v2; best = 2; }
        if (v3 > bv) { bv = v3; best = 3; }
        const int mine = (y & 1) * 2 + (x & 1);
        if (mine == best && bv > 0.f) g = gpool[plane * HP * HP + (long)py * HP + px];
    }
    gact[i] = g;
}

// the same for even H, one thread per 2x2 window (8-byte accesses, every activation read once), plus the window's
// contribution to the channel sum of gact (= the bias gradient of the convolution that produced act): part[plane *
// gridDim.x + block] = sum over the block's windows -- the caller folds them per channel (slab_reduce with inner = gridDim.x)
__global__ void __launch_bounds__(256) pool_relu_bwd2_kernel(const float* __restrict__ act, const float* __restrict__ gpool,
                                                            float* __restrict__ gact, float* __restrict__ part, int H, int HP) {
    const int wi = blockIdx.x * 256 + threadIdx.x;
    const long plane = blockIdx.y;
    float mine = 0.f;
    if (wi < HP * HP) {
        const int py = wi / HP, px = wi - py * HP;
        const float* q = act + plane * H * H + (long)(2 * py) * H + 2 * px;
        const float2 r0 = *(const float2*)q, r1 = *(const float2*)(q + H);
        int best = 0; float bv = r0.x;
        if (r0.y > bv) { bv = r0.y; best = 1; }
        if (r1.x > bv) { bv = r1.x; best = 2; }
        if (r1.y > bv) { bv = r1.y; best = 3; }
        const float g = bv > 0.f ? gpool[plane * HP * HP + wi] : 0.f;
        float* o = gact + plane * H * H + (long)(2 * py) * H + 2 * px;
        *(float2*)o = make_float2(best == 0 ? g : 0.f, best == 1 ? g : 0.f);
        *(float2*)(o + H) = make_float2(best == 2 ? g : 0.f, best == 3 ? g : 0.f);
        mine = g;
    }
    __shared__ float red[256];
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[plane * gridDim.x + blockIdx.x] = red[0];
}

__global__ void relu_mask_kernel(float* __restrict__ g, const float* __restrict__ act, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(act[i] > 0.f)) g[i] = 0.f;
}

// Fixed-order fold of partial sums: out[i] += sum_{s < nsplit} sum_{q < inner} slabs[s*stride + i*inner + q]
// (split-K weight gradients, bias-sum partials): what makes the gradients bitwise reproducible without atomics.
__global__ void __launch_bounds__(256) slab_reduce_kernel(float* __restrict__ out, const float* __restrict__ slabs, int n,
                                                         int nsplit, long stride, int inner) {
    // 16 outputs per workgroup; 16 thread groups take the slabs s = g, g+16, ... (the small bias folds have up to 256 slabs
    // and 64 outputs: with one slab chain per wave they were 15 us each, 30 of them per step) and their sums are added in
    // group order
    const int i = blockIdx.x * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    float acc = 0.f;
    if (i < n)
        for (int s = g; s < nsplit; s += 16)
            for (int q = 0; q < inner; ++q) acc += slabs[s * stride + (long)i * inner + q];
    __shared__ float red[16][17];
    red[g][threadIdx.x & 15] = acc;
    __syncthreads();
    if (g == 0 && i < n) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
        out[i] += v;
    }
}

// Up to 24 of those folds in ONE launch (the bias-gradient folds of a backward pass: 64-1536 outputs each, 2-8 workgroups,
// 5-9 us apiece as launches of their own).  Jobs are independent: different outputs, partial sums in regions of their own.
struct FoldJob { float* out; const float* slabs; long stride; int n, nsplit, inner, blk0; };
constexpr int kMaxFoldJobs = 24;
struct FoldJobs { FoldJob j[kMaxFoldJobs]; int count; };
__global__ void __launch_bounds__(256) fold_jobs_kernel(const FoldJobs J) {
    int k = 0;
    while (k + 1 < J.count && (int)blockIdx.x >= J.j[k + 1].blk0) ++k;
    const FoldJob& f = J.j[k];
    const int i = ((int)blockIdx.x - f.blk0) * 16 + (threadIdx.x & 15), g = threadIdx.x >> 4;
    float acc = 0.f;
    if (i < f.n)
        for (int s = g; s < f.nsplit; s += 16)
            for (int q = 0; q < f.inner; ++q) acc += f.slabs[s * f.stride + (long)i * f.inner + q];
    __shared__ float red[16][17];
    red[g][threadIdx.x & 15] = acc;
    __syncthreads();
    if (g == 0 && i < f.n) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][threadIdx.x];
        f.out[i] += v;
    }
}

// the same fold for a few long slabs (split-K weight gradients: n up to 786 K, 2-16 slabs): one thread per 4 outputs,
// the slabs added in order -- whole-line accesses instead of 64-byte runs, and no idle chains
__global__ void __launch_bounds__(256) slab_reduce_wide_kernel(float4* __restrict__ out, const float4* __restrict__ slabs, int n4,
                                                              int nsplit, long stride4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = out[i];
    for (int s = 0; s < nsplit; ++s) {
        const float4 p = slabs[s * stride4 + i];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    out[i] = v;
}

// part[chunk*C + c] = sum over the chunk's share of {o < outer, i < inner} of g[(o*C + c)*inner + i]  (bias gradients
// of the NCHW maps: long contiguous runs per channel)
__global__ void __launch_bounds__(256) chan_sum_kernel(const float* __restrict__ g, float* __restrict__ part, int outer,
                                                      int C, int inner) {
    const int c = blockIdx.x;
    const long per = ((long)outer * inner + gridDim.y - 1) / gridDim.y;
    const long lo = blockIdx.y * per, hi = min((long)outer * inner, lo + per);
    float acc = 0.f;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const long o = e / inner; const int i = (int)(e - o * inner);
        acc += g[(o * C + c) * inner + i];
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.y * C + c] = red[0];
}

// the same when `inner` is small (rows of cols = C*inner contiguous floats: Linear / GRU gate gradients with inner 1,
// the sequence-layout sound map with inner 7): lanes walk the row, so the reads are coalesced;
// part[chunk*cols + col] = column sums over the chunk's rows (the fold adds the `inner` columns of a channel)
__global__ void __launch_bounds__(256) col_sum_kernel(const float* __restrict__ g, float* __restrict__ part, int rows, int cols) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per, hi = min(rows, lo + per);
    float acc = 0.f;
    if (col < cols) {
        // eight independent loads in flight per thread (one at a time, the 230 MB gate-gradient maps took 150 us each),
        // added in a fixed order
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = lo + ty;
        for (; r + 28 < hi; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += g[(long)(r + 4 * u) * cols + col];
        }
        for (; r < hi; r += 4) a[0] += g[(long)r * cols + col];
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __shared__ float red[4][64];
    red[ty][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ty == 0 && col < cols)
        part[(long)blockIdx.y * cols + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// one GRU time step, both directions (torch.nn.GRU gate order r, z, n):
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h
// GI (dir, clip*T + t, 1536) holds x W_ih^T + b_ih, GH (dir, clip, 1536) holds h W_hh^T + b_hh.
// GH holds h W_hh^T WITHOUT the bias as `nsplit` split-K partial slabs, added here in fixed order.
__global__ void gru_gate_fwd_kernel(const float* __restrict__ GI, const float* __restrict__ GH, int nsplit, const float* __restrict__ hprev,
                                    float* __restrict__ hnext, float* __restrict__ R, float* __restrict__ Z,
                                    float* __restrict__ Nn, float* __restrict__ GHN, const float* __restrict__ b_hh,
                                    long dirP, int nclips, int step, long dirGI, long dirH, long dirS, int save) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nclips * kGh) return;
    const int dir = blockIdx.y;
    const int clip = i / kGh, j = i - clip * kGh;
    const int t = dir ? kSeq - 1 - step : step;
    const float* gi = GI + dir * dirGI + ((long)clip * kSeq + t) * kG3;
    const float* gh = GH + (long)dir * nclips * kG3 + (long)clip * kG3;
    const float* bh = b_hh + dir * dirP;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* q = gh + (long)sp * 2 * nclips * kG3;
        g0 += q[j]; g1 += q[kGh + j]; g2 += q[2 * kGh + j];
    }
    const float r = sigmoidf_(gi[j] + (g0 + bh[j]));
    const float z = sigmoidf_(gi[kGh + j] + (g1 + bh[kGh + j]));
    const float ghn = g2 + bh[2 * kGh + j];
    const float n = tanhf(gi[2 * kGh + j] + r * ghn);
    const float hp = hprev[dir * dirH + i];
    hnext[dir * dirH + i] = (1.f - z) * n + z * hp;
    if (save) {
        const long o = dir * dirS + (long)step * nclips * kGh + i;
        R[o] = r; Z[o] = z; Nn[o] = n; GHN[o] = ghn;
    }
}

// backward of that step: from dh (in place -> dh * z, the direct path to h_prev) to the gate pre-activation
// gradients, DGI in (dir, clip*T + t, 1536) and DGH in (dir, step, clip, 1536).
__global__ void gru_gate_bwd_kernel(float* __restrict__ DH, const float* __restrict__ DHP, int nparts,
                                    const float* __restrict__ hprev, const float* __restrict__ R,
                                    const float* __restrict__ Z, const float* __restrict__ Nn, const float* __restrict__ GHN,
                                    float* __restrict__ DGI, float* __restrict__ DGH, int nclips, int step, long dirGI,
                                    long dirH, long dirS, long dirDGH) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nclips * kGh) return;
    const int dir = blockIdx.y;
    const int clip = i / kGh, j = i - clip * kGh;
    const int t = dir ? kSeq - 1 - step : step;
    const long o = dir * dirS + (long)step * nclips * kGh + i;
    const float r = R[o], z = Z[o], n = Nn[o], ghn = GHN[o];
    float dh = DH[(long)dir * nclips * kGh + i];          // direct path (dh * z of the later step, or the head's gradient)
    for (int sp = 0; sp < nparts; ++sp) dh += DHP[(long)sp * 2 * nclips * kGh + (long)dir * nclips * kGh + i];   // + dgh W_hh
    const float hp = hprev[dir * dirH + i];
    const float dn_pre = dh * (1.f - z) * (1.f - n * n);
    const float dz_pre = dh * (hp - n) * z * (1.f - z);
    const float dr_pre = dn_pre * ghn * r * (1.f - r);
    float* gi = DGI + dir * dirGI + ((long)clip * kSeq + t) * kG3;
    float* gh = DGH + dir * dirDGH + ((long)step * nclips + clip) * kG3;
    gi[j] = dr_pre; gi[kGh + j] = dz_pre; gi[2 * kGh + j] = dn_pre;
    gh[j] = dr_pre; gh[kGh + j] = dz_pre; gh[2 * kGh + j] = dn_pre * r;
    DH[(long)dir * nclips * kGh + i] = dh * z;
}

// sraw[clip] = [h_T forward | h_T reverse]; and its inverse for the gradient
__global__ void gru_concat_kernel(const float* __restrict__ hfin, float* __restrict__ sraw, int nclips, long dirH, int inverse) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nclips * kSRaw) return;
    const int clip = i / kSRaw, q = i - clip * kSRaw, dir = q / kGh, j = q - dir * kGh;
    if (inverse) ((float*)hfin)[dir * dirH + (long)clip * kGh + j] = sraw[i];
    else sraw[i] = hfin[dir * dirH + (long)clip * kGh + j];
}

// F.normalize(x, p=2, dim=1) on rows of 3 (pretext_base.py:18,23) and its backward
__global__ void l2norm_fwd_kernel(const float* __restrict__ raw, float* __restrict__ emb, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float x = raw[3 * i], y = raw[3 * i + 1], z = raw[3 * i + 2];
    const float nrm = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    emb[3 * i] = x / nrm; emb[3 * i + 1] = y / nrm; emb[3 * i + 2] = z / nrm;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ gemb, float* __restrict__ graw, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float x = raw[3 * i], y = raw[3 * i + 1], z = raw[3 * i + 2];
    const float gx = gemb[3 * i], gy = gemb[3 * i + 1], gz = gemb[3 * i + 2];
    const float n2 = sqrtf(x * x + y * y + z * z);
    if (n2 < 1e-12f) { graw[3 * i] = gx / 1e-12f; graw[3 * i + 1] = gy / 1e-12f; graw[3 * i + 2] = gz / 1e-12f; return; }
    const float ex = x / n2, ey = y / n2, ez = z / n2;
    const float dot = gx * ex + gy * ey + gz * ez;
    graw[3 * i] = (gx - ex * dot) / n2; graw[3 * i + 1] = (gy - ey * dot) / n2; graw[3 * i + 2] = (gz - ez * dot) / n2;
}

__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- launch helpers ---------------------------------------------------------------------------------------------
static inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

#define IT_CHECK(c) VAR_HIP_CHECK(c, hipGetLastError())

template <class G, bool U8, bool SEQ>
static int conv_fwd(var_ctx* c, hipStream_t s, const ConvDims& d, const void* x, const float* w, const float* bias, float* y) {
    ConvFwdP<G, U8, SEQ> p{};
    p.M = d.B * d.HO * d.WO; p.N = d.COUT; p.K = d.CIN * G::KHW; p.nsplit = 1;
    p.d = d; p.x = x; p.w = w; p.bias = bias; p.y = y;
    return gg(c, s, p);
}
template <class G, bool SEQ>
static int conv_dgrad(var_ctx* c, hipStream_t s, const ConvDims& d, const float* gy, const float* w, float* dx,
                      const float* mask = nullptr) {
    if constexpr (G::SH == 2 && G::SW == 2) {
        ConvDgradS2P<G, SEQ> p{};
        p.H2 = (d.H + 1) / 2; p.W2 = (d.W + 1) / 2;
        p.inv_h2w2 = 1.f / (float)(p.H2 * p.W2); p.inv_w2 = 1.f / (float)p.W2;
        p.M = d.B * p.H2 * p.W2; p.N = d.CIN; p.K = d.COUT * ConvDgradS2P<G, SEQ>::NKY * ConvDgradS2P<G, SEQ>::NKX; p.nsplit = 1;
        p.d = d; p.gy = gy; p.w = w; p.dx = dx; p.mask = mask;
        return gg(c, s, p, 4);
    } else {
        static_assert(!SEQ, "sequence layout only on the stride-2 sound layer");
        ConvDgradP<G> p{};
        p.M = d.B * d.H * d.W; p.N = d.CIN; p.K = d.COUT * G::KHW; p.nsplit = 1;
        p.d = d; p.gy = gy; p.w = w; p.dx = dx; p.mask = mask;
        return gg(c, s, p);
    }
}
static constexpr long kSlabFloats = 24L << 20;        // split-K slabs (96 MB)
static constexpr long kBiasSlabFloats = 4L << 20;     // bias-sum partials of a whole backward pass (bs_take)
static constexpr long kPartFloats = 1L << 18;          // room for a kernel that reports its partial count after the launch

// the kernel gives split s the chunks [s*per, (s+1)*per), per = ceil(chunks / nsplit): trim nsplit so that no split
// is empty (an empty split would leave its slab unwritten)
static int eff_split(int K, int ns, int kc = GG_KC) {
    const int kchunks = (K + kc - 1) / kc;
    if (ns < 1) ns = 1;
    if (ns > kchunks) ns = kchunks;
    const int per = (kchunks + ns - 1) / ns;
    return (kchunks + per - 1) / per;
}

// K splits of a small recurrent product so that about two workgroups per CU are in flight
static int rec_split(int tiles, int kchunks, int cap) {
    int ns = (256 + tiles - 1) / tiles;        // the consumer reads every partial slab: keep the count small
    if (ns > cap) ns = cap;
    if (ns > kchunks / 4) ns = kchunks / 4;
    return eff_split(kchunks * GG_KC, ns);
}

static int flush_folds(var_ctx* c, hipStream_t s) {
    FoldJobs* J = (FoldJobs*)ith(c)->folds;
    if (!J || !J->count) return VAR_OK;
    const FoldJob& last = J->j[J->count - 1];
    const int blocks = last.blk0 + (last.n + 15) / 16;
    hipLaunchKernelGGL(fold_jobs_kernel, dim3(blocks), dim3(256), 0, s, *J);
    J->count = 0;
    IT_CHECK(c);
    return VAR_OK;
}
// n floats of the bias-partial buffer that stay untouched until the next flush_folds
static float* bs_take(var_ctx* c, hipStream_t s, long n) {
    ithor_state* st = ith(c);
    n = (n + 63) & ~63L;
    if (st->bs_used + n > kBiasSlabFloats) { (void)flush_folds(c, s); st->bs_used = 0; }
    float* p = st->bslab + st->bs_used;
    st->bs_used += n;
    return p;
}
static int slab_reduce(var_ctx* c, hipStream_t s, float* out, const float* slabs, int n, int nsplit, long stride, int inner = 1) {
    ithor_state* st = ith(c);
    if (st->folds && slabs >= st->bslab && slabs < st->bslab + kBiasSlabFloats) {     // partials from bs_take: fold later, together
        FoldJobs* J = (FoldJobs*)st->folds;
        for (int k = 0; k < J->count; ++k)                   // (two folds into the same outputs keep their order)
            if (out < J->j[k].out + J->j[k].n && J->j[k].out < out + n) { const int r = flush_folds(c, s); if (r != VAR_OK) return r; break; }
        if (J->count == kMaxFoldJobs) { const int r = flush_folds(c, s); if (r != VAR_OK) return r; }
        FoldJob& f = J->j[J->count];
        f.out = out; f.slabs = slabs; f.stride = stride; f.n = n; f.nsplit = nsplit; f.inner = inner;
        f.blk0 = J->count ? J->j[J->count - 1].blk0 + (J->j[J->count - 1].n + 15) / 16 : 0;
        J->count++;
        return VAR_OK;
    }
    if (inner == 1 && nsplit <= 16 && n >= 4096 && n % 4 == 0 && stride % 4 == 0 && (((uintptr_t)out | (uintptr_t)slabs) & 15) == 0) {
        hipLaunchKernelGGL(slab_reduce_wide_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, s, (float4*)out, (const float4*)slabs, n / 4,
                           nsplit, stride / 4);
        IT_CHECK(c);
        return VAR_OK;
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, s, out, slabs, n, nsplit, stride, inner);
    IT_CHECK(c);
    return VAR_OK;
}

template <class G, bool U8, bool SEQ>
static int conv_wgrad(var_ctx* c, hipStream_t s, const ConvDims& d, const void* x, const float* gy, float* dw) {
    ConvWgradP<G, U8, SEQ> p{};
    p.M = d.CIN * G::KHW; p.N = d.COUT; p.K = d.B * d.HO * d.WO;
    const int tiles = ((p.M + GG_MT - 1) / GG_MT) * ((p.N + 63) / 64);
    int ns = (1024 + tiles - 1) / tiles;                    // about 4 workgroups per CU
    // both operands are read along k (pixels): chunks of 32 k make the runs per lane group twice as long
    constexpr int KC = 32;
    const int kchunks = (p.K + KC - 1) / KC;
    if (ns > kchunks / 4) ns = kchunks / 4 > 0 ? kchunks / 4 : 1;
    if ((long)ns * p.M * p.N > kSlabFloats) ns = (int)(kSlabFloats / ((long)p.M * p.N));
    p.nsplit = eff_split(p.K, ns, KC);
    p.d = d; p.x = x; p.gy = gy; p.dw = dw; p.slab = ith(c)->slab;
    int r = gg<KC>(c, s, p);
    if (r != VAR_OK) return r;
    if (p.nsplit > 1) return slab_reduce(c, s, dw, p.slab, p.M * p.N, p.nsplit, (long)p.M * p.N);
    return VAR_OK;
}
static int chan_sum(var_ctx* c, hipStream_t s, const float* g, float* out, int outer, int C, int inner) {
    if (inner < 64) {
        const int cols = C * inner;
        int chunks = (outer + 255) / 256;
        if (chunks > 32) chunks = 32;
        float* part = bs_take(c, s, (long)chunks * cols);
        hipLaunchKernelGGL(col_sum_kernel, dim3((cols + 63) / 64, chunks), dim3(256), 0, s, g, part, outer, cols);
        IT_CHECK(c);
        return slab_reduce(c, s, out, part, C, chunks, cols, inner);
    }
    long tot = (long)outer * inner;
    int chunks = (int)((tot + 8191) / 8192);
    if (chunks > 32) chunks = 32;
    if (chunks < 1) chunks = 1;
    float* part = bs_take(c, s, (long)chunks * C);
    hipLaunchKernelGGL(chan_sum_kernel, dim3(C, chunks), dim3(256), 0, s, g, part, outer, C, inner);
    IT_CHECK(c);
    return slab_reduce(c, s, out, part, C, chunks, C);
}

// ---- the heads' small Linear layers in bf16 mode ---------------------------------------------------------------------
// 128 -> 3, 128 -> 64, 64 -> 3, 1024 -> 128 over a few hundred rows: one or a handful of GEMM tiles each, 9-26 us apiece
// on either GEMM engine (pure launch / pipeline latency), eleven products per step.  Plain vector-ALU kernels instead:
// operands rounded to bf16 like every product of the mode, fp32 accumulation in a fixed order.
__device__ __forceinline__ float bf16r(float x) { return __uint_as_float(dense16::bf16_bits(x) << 16); }
// Y[row][o] = act(b[o] + sum_k X[row][k] W[o][k]): one wave per (row, o), lanes stride K (both rows read coalesced)
__global__ void __launch_bounds__(256) small_linear_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                               const float* __restrict__ b, float* __restrict__ Y, int rows, int K, int O,
                                                               int relu) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= rows * O) return;
    const int row = pair / O, o = pair - row * O;
    const float* x = X + (long)row * K;
    const float* w = W + (long)o * K;
    float acc = 0.f;
    int k = lane;
    for (; k + 64 * 7 < K; k += 64 * 8) {
        float a[8], bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = x[k + 64 * u]; bb[u] = w[k + 64 * u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += bf16r(a[u]) * bf16r(bb[u]);
    }
    for (; k < K; k += 64) acc += bf16r(x[k]) * bf16r(w[k]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) { float v = acc + (b ? b[o] : 0.f); Y[pair] = relu ? fmaxf(v, 0.f) : v; }
}
// dW[o][k] = sum_row dY[row][o] X[row][k]: one thread per (o, k), consecutive threads walk k
// (and db[o] = sum_row dY[row][o], unrounded, by the thread of k = 0: it reads that column anyway)
__global__ void __launch_bounds__(256) small_linear_dw_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dW,
                                                              float* __restrict__ db, int rows, int K, int O) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= O * K) return;
    const int o = i / K, k = i - o * K;
    float acc = 0.f, gs = 0.f;
    int r = 0;
    for (; r + 16 <= rows; r += 16) {                        // 32 loads in flight, then 16 products in row order
        float g[16], x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { g[u] = dY[(long)(r + u) * O + o]; x[u] = X[(long)(r + u) * K + k]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc += bf16r(g[u]) * bf16r(x[u]); gs += g[u]; }
    }
    for (; r < rows; ++r) { const float g = dY[(long)r * O + o]; acc += bf16r(g) * bf16r(X[(long)r * K + k]); gs += g; }
    dW[i] = acc;
    if (k == 0) db[o] = gs;
}
// dX[row][k] = sum_o dY[row][o] W[o][k]: one thread per (row, k)
__global__ void __launch_bounds__(256) small_linear_dx_kernel(const float* __restrict__ W, const float* __restrict__ dY, float* __restrict__ dX,
                                                              int rows, int K, int O) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * K) return;
    const int row = i / K, k = i - row * K;
    float acc = 0.f;
    int o = 0;
    for (; o + 16 <= O; o += 16) {
        float g[16], w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { g[u] = dY[(long)row * O + o + u]; w[u] = W[(long)(o + u) * K + k]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += bf16r(g[u]) * bf16r(w[u]);
    }
    for (; o < O; ++o) acc += bf16r(dY[(long)row * O + o]) * bf16r(W[(long)o * K + k]);
    dX[i] = acc;
}
static inline bool small_linear(const ithor_state* st, int rows, int K, int O) { return st->bf16 && (long)rows * K * O <= (80L << 20); }

// Y (rows, O) = X (rows, K) W^T + b, optional ReLU
static int linear_fwd(var_ctx* c, hipStream_t s, const float* X, const float* W, const float* b, float* Y, int rows, int K,
                      int O, int relu) {
    DenseP<true, true, 0> p{};
    p.M = O; p.N = rows; p.K = K; p.nsplit = 1;
    p.A = W; p.sam = K; p.sak = 1; p.Bm = X; p.sbk = 1; p.sbn = K; p.C = Y; p.scm = 1; p.scn = O; p.bias = b; p.relu = relu;
    ithor_state* st = ith(c);
    const int tiles = ((O + dense16::TM - 1) / dense16::TM) * ((rows + dense16::TN - 1) / dense16::TN);
    if (st->bf16 && dense16::eligible(p) && tiles <= 16 && K >= 512) {
        // a handful of tiles with a long K (the image head's 1152 -> 128 at batch 256 is TWO tiles: 67 us): split K over the
        // grid into slabs, add them with bias and activation in a finish pass
        DenseP<true, true, 2> q{};
        q.M = O; q.N = rows; q.K = K; q.nsplit = eff_split(K, 128 / tiles > 16 ? 16 : 128 / tiles, dense16::TK);
        q.A = W; q.sam = K; q.sak = 1; q.Bm = X; q.sbk = 1; q.sbn = K; q.C = st->slab; q.scm = 1; q.scn = O; q.sC = (long)rows * O;
        if (q.nsplit > 1 && q.nsplit * q.sC <= kSlabFloats && dense16::eligible(q)) {
            int r = gg(c, s, q); if (r) return r;
            const long n = (long)rows * O;
            hipLaunchKernelGGL(gg_finish_kernel, g1(n), dim3(256), 0, s, Y, st->slab, n, q.nsplit, q.sC, b, O, 1, relu);
            IT_CHECK(c);
            return VAR_OK;
        }
    }
    if (small_linear(st, rows, K, O)) {
        hipLaunchKernelGGL(small_linear_fwd_kernel, dim3((rows * O + 3) / 4), dim3(256), 0, s, X, W, b, Y, rows, K, O, relu);
        IT_CHECK(c);
        return VAR_OK;
    }
    return gg(c, s, p);
}
// backward of that layer from dY (already masked by the layer's own ReLU): dW += dY^T X, db += colsum(dY), dX = dY W
static int linear_bwd(var_ctx* c, hipStream_t s, const float* X, const float* W, const float* dY, float* dW, float* db,
                      float* dX, int rows, int K, int O) {
    const bool small = small_linear(ith(c), rows, K, O);
    if (small) {
        hipLaunchKernelGGL(small_linear_dw_kernel, g1((long)O * K), dim3(256), 0, s, X, dY, dW, db, rows, K, O);
        IT_CHECK(c);
    } else {
        DenseP<false, false, 0> p{};
        p.M = K; p.N = O; p.K = rows; p.nsplit = 1;
        p.A = X; p.sam = 1; p.sak = K; p.Bm = dY; p.sbk = O; p.sbn = 1; p.C = dW; p.scm = 1; p.scn = K;
        int r = gg(c, s, p); if (r) return r;
    }
    int r = small ? VAR_OK : chan_sum(c, s, dY, db, rows, O, 1); if (r) return r;
    if (dX && small) {
        hipLaunchKernelGGL(small_linear_dx_kernel, g1((long)rows * K), dim3(256), 0, s, W, dY, dX, rows, K, O);
        IT_CHECK(c);
    } else if (dX) {
        DenseP<false, true, 0> p{};
        p.M = K; p.N = rows; p.K = O; p.nsplit = 1;
        p.A = W; p.sam = 1; p.sak = K; p.Bm = dY; p.sbk = 1; p.sbn = O; p.C = dX; p.scm = 1; p.scn = K;
        r = gg(c, s, p); if (r) return r;
    }
    return VAR_OK;
}
static int relu_mask(var_ctx* c, hipStream_t s, float* g, const float* act, long n) {
    hipLaunchKernelGGL(relu_mask_kernel, g1(n), dim3(256), 0, s, g, act, n);
    IT_CHECK(c);
    return VAR_OK;
}

using G3s1 = Geo<3, 3, 1, 1, 1, 1>;
using G3s2 = Geo<3, 3, 2, 2, 1, 1>;
using GS1 = Geo<11, 11, 2, 2, 5, 5>;
using GS2 = Geo<11, 5, 2, 2, 5, 5>;
using GS3 = Geo<7, 3, 2, 2, 1, 1>;

static ConvDims img_dims(const ithor_state* st, int l, int B) {      // l = 1..6
    const int hin = l == 1 ? st->hs[0] : (l == 2 ? st->hs[0] : st->hs[l - 2]);
    return conv_dims(B, kICh[l - 1], hin, hin, kICh[l], 3, 3, l == 6 ? 2 : 1, l == 6 ? 2 : 1, 1, 1);
}
static ConvDims snd_dims(int l, int n) {                              // l = 1..3
    if (l == 1) return conv_dims(n, 1, kT, kF, 64, 11, 11, 2, 2, 5, 5);
    if (l == 2) return conv_dims(n, 64, 300, 20, 64, 11, 5, 2, 2, 5, 5);
    return conv_dims(n, 64, 150, 13, 64, 7, 3, 2, 2, 1, 1);
}

#define RUN(x) do { int r_ = (x); if (r_ != VAR_OK) return r_; } while (0)

// ---- layer 6 (3x3 stride 2 on a 6 x 6 | 5 x 5 map -> 3 x 3) as a dense layer, bf16 mode -----------------------------------
// On maps this small the gather-GEMM spends 93 us on 0.7 GFLOP (and 46 us on the data gradient).  The convolution IS a
// matrix product with the image's whole input map as the reduction index: out[b][oc*9 + p] = sum_k W'[oc*9 + p][k] x[b][k],
// k = c*HP*HP + iy*HP + ix, W' = the filter scattered to the positions each output pixel sees (zeros elsewhere: 4x the
// arithmetic, still nothing).  W' is rebuilt from the parameters in every forward (10 MB of bf16); both products then run
// on the dense kernel like every other big product of the mode, the data gradient with W' transposed by its strides.
__global__ void __launch_bounds__(256) l6_expand_kernel(const float* __restrict__ w, const float* __restrict__ bias, uint4* __restrict__ wx,
                                                        float* __restrict__ bx, int HP) {
    const int KX = 128 * HP * HP, K8 = KX / 8;
    const int i = blockIdx.x * 256 + threadIdx.x;               // (m, 8 consecutive k)
    if (i < kIRaw) bx[i] = bias[i / 9];
    if (i >= kIRaw * K8) return;
    const int m = i / K8, k0 = (i - m * K8) * 8, oc = m / 9, p = m - oc * 9, oy = p / 3, ox = p - oy * 3;
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e, c = k / (HP * HP), r = k - c * HP * HP, iy = r / HP, ix = r - iy * HP;
        const int ky = iy - (2 * oy - 1), kx = ix - (2 * ox - 1);
        const bool hit = (unsigned)ky < 3u && (unsigned)kx < 3u;
        const float t = w[((oc * 128 + c) * 3 + (hit ? ky : 0)) * 3 + (hit ? kx : 0)];
        v[e] = hit ? dense16::bf16_bits(t) : 0u;
    }
    wx[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// 0 = done on the dense kernel, 1 = take the convolution path (fp32 mode, batch < 64, a map the expansion does not cover)
static int l6_dense_fwd(var_ctx* c, hipStream_t s, const float* P, int B) {
    ithor_state* st = ith(c);
    const IthorLayout& L = st->L;
    const int HP = st->hs[4], KX = 128 * HP * HP;
    st->l6_ready = false;
    if (!st->bf16 || B < 64 || st->hs[5] != 3 || (HP != 5 && HP != 6) || !st->l6w) return 1;
    hipLaunchKernelGGL(l6_expand_kernel, g1((long)kIRaw * (KX / 8)), dim3(256), 0, s, P + L.iw[5], P + L.ib[5], (uint4*)st->l6w, st->l6b, HP);
    IT_CHECK(c);
    RUN(gru_bf16_to_bf16(c, s, st->p[5], st->l6x, (long)B * KX));
    // 9 x 2 tiles only: K is split 8 ways over the grid (slabs), a finish pass adds them with bias and ReLU
    DenseP<true, true, 2> p{};
    p.M = kIRaw; p.N = B; p.K = KX; p.nsplit = eff_split(KX, 8, dense16::TK);
    p.A = (const float*)st->l6w; p.sam = KX; p.sak = 1; p.a16 = 1;
    p.Bm = (const float*)st->l6x; p.sbk = 1; p.sbn = KX; p.b16 = 1;
    p.C = st->slab; p.scm = 1; p.scn = kIRaw; p.sC = (long)B * kIRaw;
    if (!dense16::eligible(p) || p.nsplit * p.sC > kSlabFloats) return 1;
    RUN(gg(c, s, p));
    const long n = (long)B * kIRaw;
    hipLaunchKernelGGL(gg_finish_kernel, g1(n), dim3(256), 0, s, st->a[6], st->slab, n, p.nsplit, p.sC, P + L.ib[5], 128, 9, 1);
    IT_CHECK(c);
    st->l6_ready = true;
    return VAR_OK;
}
// gp[5] = ga[6] W'  (after l6_dense_fwd of the same step)
static int l6_dense_dgrad(var_ctx* c, hipStream_t s, int B) {
    ithor_state* st = ith(c);
    const int HP = st->hs[4], KX = 128 * HP * HP;
    if (!st->bf16 || !st->l6_ready) return 1;
    RUN(gru_bf16_to_bf16(c, s, st->ga[6], st->l6g, (long)B * kIRaw));
    DenseP<false, true, 2> p{};
    p.M = KX; p.N = B; p.K = kIRaw; p.nsplit = eff_split(kIRaw, 4, dense16::TK);
    p.A = (const float*)st->l6w; p.sam = 1; p.sak = KX; p.a16 = 1;
    p.Bm = (const float*)st->l6g; p.sbk = 1; p.sbn = kIRaw; p.b16 = 1;
    p.C = st->slab; p.scm = 1; p.scn = KX; p.sC = (long)B * KX;
    if (!dense16::eligible(p) || p.nsplit * p.sC > kSlabFloats) return 1;
    RUN(gg(c, s, p));
    const long n = (long)B * KX;
    hipLaunchKernelGGL(gg_finish_kernel, g1(n), dim3(256), 0, s, st->gp[5], st->slab, n, p.nsplit, p.sC, (const float*)nullptr, 1, 1, 0);
    IT_CHECK(c);
    return VAR_OK;
}

// ---- forward ----------------------------------------------------------------------------------------------------
static int ithor_fwd(var_ctx* c, hipStream_t s, const float* P, const void* image, int is_u8, long bstride, const float* pos,
                     const float* neg, int B, bool save) {
    ithor_state* st = ith(c);
    const IthorLayout& L = st->L;
    const int* hs = st->hs;
    st->B = B; st->gen = ++c->fwd_gen; st->has_img = image != nullptr; st->has_pos = pos != nullptr; st->has_neg = neg != nullptr;
    st->image = image; st->is_u8 = is_u8; st->bstride = bstride; st->pos = pos; st->neg = neg;
    const int nclips = (pos ? B : 0) + (neg ? B : 0);
    st->nclips = nclips;
    // a time-out of the PREVIOUS step stops guarding Adam here (it stays visible in var_ithor_gru_status' sticky words)
    if (st->bf16 && st->gru_seq) RUN(gru_bf16_step_begin(c, s, 2 * st->maxB, st->gruws));
    if (image) {
        st->img_packed = false;
        if (st->bf16 && hs[0] == 96) {      // the eight filter tables of layers 2-5 (forward and data gradient) in one launch
            RUN(img_bf16_pack_all(c, s, P + L.iw[1], P + L.iw[2], P + L.iw[3], P + L.iw[4], st->imgws));
            st->img_packed = true;
        }
        ConvDims d = img_dims(st, 1, B);
        d.xb = bstride;
        if (is_u8) RUN((conv_fwd<G3s1, true, false>(c, s, d, image, P + L.iw[0], P + L.ib[0], st->a[1])));
        else RUN((conv_fwd<G3s1, false, false>(c, s, d, image, P + L.iw[0], P + L.ib[0], st->a[1])));
        {
            int r = st->bf16 ? img_bf16_conv(c, s, 2, hs[0], 0, st->a[1], P + L.iw[1], P + L.ib[1], nullptr, st->a[2], nullptr, nullptr, B, st->imgws, st->img_packed ? 1 : 0) : 1;
            if (r == 1) r = conv_fwd<G3s1, false, false>(c, s, img_dims(st, 2, B), st->a[1], P + L.iw[1], P + L.ib[1], st->a[2]);
            RUN(r);
        }
        for (int l = 2; l <= 5; ++l) {
            // pool the output of conv l into p[l], then conv l+1
            const long n = (long)B * kICh[l] * hs[l - 1] * hs[l - 1];
            const int hin = l == 2 ? hs[0] : hs[l - 2];
            hipLaunchKernelGGL(pool_fwd_kernel, g1(n), dim3(256), 0, s, st->a[l], st->p[l], n, hin, hs[l - 1]);
            IT_CHECK(c);
            if (l < 5) {
                int r = st->bf16 ? img_bf16_conv(c, s, l + 1, hs[l - 1], 0, st->p[l], P + L.iw[l], P + L.ib[l], nullptr, st->a[l + 1], nullptr, nullptr, B, st->imgws, st->img_packed ? 1 : 0) : 1;
                if (r == 1) r = conv_fwd<G3s1, false, false>(c, s, img_dims(st, l + 1, B), st->p[l], P + L.iw[l], P + L.ib[l], st->a[l + 1]);
                RUN(r);
            }
            else {
                int r = l6_dense_fwd(c, s, P, B);
                if (r == 1) r = conv_fwd<G3s2, false, false>(c, s, img_dims(st, 6, B), st->p[5], P + L.iw[5], P + L.ib[5], st->a[6]);
                RUN(r);
            }
        }
        RUN(linear_fwd(c, s, st->a[6], P + L.ih_w0, P + L.ih_b0, st->hid_i, B, kIRaw, 128, 1));
        RUN(linear_fwd(c, s, st->hid_i, P + L.ih_w1, P + L.ih_b1, st->raw, B, 128, 3, 0));
    }
    if (nclips) {
        // clips: [pos | neg] (whichever are given), local index 0..nclips-1
        if (st->bf16) {      // both sounds in one launch: two clips per CU
            RUN(snd1_bf16_fwd(c, s, pos ? pos : neg, B, pos && neg ? neg : nullptr, pos && neg ? B : 0, P + L.sw[0], P + L.sb[0],
                              st->keep32 ? st->s[1] : nullptr, 2 * st->maxB, st->bfws));
        } else {
            int off = 0;
            for (int q = 0; q < 2; ++q) {
                const float* src = q == 0 ? pos : neg;
                if (!src) continue;
                RUN((conv_fwd<GS1, false, false>(c, s, snd_dims(1, B), src, P + L.sw[0], P + L.sb[0],
                                                 st->s[1] + (long)off * 64 * 300 * 20)));
                off += B;
            }
        }
        if (st->bf16) {      // (reads conv 1's C8 image; times its main kernel under the same tag itself)
            RUN(snd2_bf16_fwd(c, s, nullptr, P + L.sw[1], P + L.sb[1], st->keep32 ? st->s[2] : nullptr, nclips, 2 * st->maxB, st->bfws));
        } else {
            ProfScope prof(c, s, TAG_ITHOR_S2_FWD);
            RUN((conv_fwd<GS2, false, false>(c, s, snd_dims(2, nclips), st->s[1], P + L.sw[1], P + L.sb[1], st->s[2])));
        }
        if (st->bf16) RUN(snd3_bf16_fwd(c, s, P + L.sw[2], P + L.sb[2], st->s[3], nclips, 2 * st->maxB, st->bfws));
        else RUN((conv_fwd<GS3, false, true>(c, s, snd_dims(3, nclips), st->s[2], P + L.sw[2], P + L.sb[2], st->s[3])));
        const int rows = nclips * kSeq;
        const long dirP = L.w_ih[1] - L.w_ih[0];
        const long dirGI = (long)rows * kG3, dirH = (long)(kSeq + 1) * nclips * kGh, dirS = (long)kSeq * nclips * kGh;
        {
            DenseP<true, true, 0> p{};
            p.M = kG3; p.N = rows; p.K = kGin; p.nsplit = 1;
            p.A = P + L.w_ih[0]; p.sam = kGin; p.sak = 1; p.zA = dirP;
            p.Bm = st->s[3]; p.sbk = 1; p.sbn = kGin; p.zB = 0;
            p.C = st->GI; p.scm = 1; p.scn = kG3; p.zC = dirGI; p.bias = P + L.b_ih[0]; p.zbias = dirP;
            if (st->bf16) {
                RUN(gru_bf16_pack(c, s, P + L.w_hh[0], P + L.w_ih[0], dirP, nclips, 2 * st->maxB, st->Hb, dirH, st->gruws));
                RUN(gru_bf16_convert_x(c, s, st->s[3], (long)rows * kGin, 2 * st->maxB, st->gruws));
                auto q = p;
                q.zA = (long)kG3 * kGin;
                use_bf16_copies(c, q, gru_bf16_wih16(st->gruws, 2 * st->maxB), gru_bf16_x16(st->gruws, 2 * st->maxB));
                if (q.a16) p = q; else use_bf16_copies(c, p, nullptr, gru_bf16_x16(st->gruws, 2 * st->maxB));
            }
            RUN(gg(c, s, p, 2));
        }
        if (!st->bf16)      // (bf16 mode: gru_bf16_pack zeroed the initial states)
            for (int d = 0; d < 2; ++d) RUN(var_zero_async(c, s, st->Hb + d * dirH, sizeof(float) * nclips * kGh));
        int whole = 0;       // the 73 steps in one launch
        if (st->bf16 && st->gru_seq) {
            const int r = gru_bf16_seq_fwd(c, s, st->GI, st->Hb, P + L.b_hh[0], dirP, st->R, st->Z, st->Nn, st->GHN, nclips, 2 * st->maxB,
                                           dirGI, dirH, dirS, save ? 1 : 0, st->gruws, st->gru_drop_one ? 1 : 0);
            st->gru_drop_one = false;
            if (r < 0) return r;
            whole = r == 0;
        }
        for (int step = 0; step < kSeq && !whole; ++step) {
            if (st->bf16) {      // product + gates in one launch (gru_bf16.hip)
                RUN(gru_bf16_step_fwd(c, s, st->GI, st->Hb, P + L.b_hh[0], dirP, st->R, st->Z, st->Nn, st->GHN, nclips, 2 * st->maxB, step,
                                      dirGI, dirH, dirS, save ? 1 : 0, st->gruws));
                continue;
            }
            // split over K into partial slabs of GH that the gate kernel adds in fixed order
            DenseP<true, true, 2> p{};
            p.M = kG3; p.N = nclips; p.K = kGh;
            p.A = P + L.w_hh[0]; p.sam = kGh; p.sak = 1; p.zA = dirP;
            p.Bm = st->Hb + (long)step * nclips * kGh; p.sbk = 1; p.sbn = kGh; p.zB = dirH;
            p.C = st->GH; p.scm = 1; p.scn = kG3; p.zC = (long)nclips * kG3; p.sC = 2L * nclips * kG3;
            p.nsplit = rec_split(12 * ((nclips + 63) / 64) * 2, kGh / GG_KC, 4);
            RUN(gg(c, s, p, 2));
            hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3((nclips * kGh + 255) / 256, 2), dim3(256), 0, s, st->GI, st->GH, p.nsplit,
                               st->Hb + (long)step * nclips * kGh, st->Hb + (long)(step + 1) * nclips * kGh, st->R, st->Z,
                               st->Nn, st->GHN, P + L.b_hh[0], dirP, nclips, step, dirGI, dirH, dirS, save ? 1 : 0);
            IT_CHECK(c);
        }
        hipLaunchKernelGGL(gru_concat_kernel, g1((long)nclips * kSRaw), dim3(256), 0, s,
                           st->Hb + (long)kSeq * nclips * kGh, st->sraw, nclips, dirH, 0);
        IT_CHECK(c);
        RUN(linear_fwd(c, s, st->sraw, P + L.sh_w0, P + L.sh_b0, st->hid_s1, nclips, kSRaw, 128, 1));
        RUN(linear_fwd(c, s, st->hid_s1, P + L.sh_w1, P + L.sh_b1, st->hid_s2, nclips, 128, 64, 1));
        RUN(linear_fwd(c, s, st->hid_s2, P + L.sh_w2, P + L.sh_b2, st->raw + 3 * (long)st->maxB, nclips, 64, 3, 0));
    }
    if (image) { hipLaunchKernelGGL(l2norm_fwd_kernel, g1(B), dim3(256), 0, s, st->raw, st->emb, B); IT_CHECK(c); }
    if (nclips) {
        hipLaunchKernelGGL(l2norm_fwd_kernel, g1(nclips), dim3(256), 0, s, st->raw + 3 * (long)st->maxB,
                           st->emb + 3 * (long)st->maxB, nclips);
        IT_CHECK(c);
        if (st->bf16 && st->gru_seq) RUN(gru_bf16_poison_on_timeout(c, s, st->emb + 3 * (long)st->maxB, 3 * nclips, 2 * st->maxB, st->gruws));
    }
    return VAR_OK;
}

// ---- backward from gemb (gradients wrt the normalised embeddings, rows [img | clips]) ---------------------------
static int ithor_bwd(var_ctx* c, hipStream_t s, const float* P, float* G) {
    ithor_state* st = ith(c);
    const IthorLayout& L = st->L;
    const int* hs = st->hs;
    const int B = st->B, nclips = st->nclips;
    const long mB = st->maxB;
    RUN(var_zero_async(c, s, G, sizeof(float) * L.total));
    st->bs_used = 0;                                           // the pass's bias partials and their folds (flush_folds at its end)
    if (!st->folds) st->folds = calloc(1, sizeof(FoldJobs));
    if (st->folds) ((FoldJobs*)st->folds)->count = 0;
    if (st->has_img) {
        hipLaunchKernelGGL(l2norm_bwd_kernel, g1(B), dim3(256), 0, s, st->raw, st->gemb, st->graw, B);
        IT_CHECK(c);
        RUN(linear_bwd(c, s, st->hid_i, P + L.ih_w1, st->graw, G + L.ih_w1, G + L.ih_b1, st->ghid_i, B, 128, 3));
        RUN(relu_mask(c, s, st->ghid_i, st->hid_i, (long)B * 128));
        RUN(linear_bwd(c, s, st->a[6], P + L.ih_w0, st->ghid_i, G + L.ih_w0, G + L.ih_b0, st->ga[6], B, kIRaw, 128));
        RUN(relu_mask(c, s, st->ga[6], st->a[6], (long)B * kIRaw));
        // conv 6 (stride 2) on p[5]
        {
            const ConvDims d = img_dims(st, 6, B);
            RUN((conv_wgrad<G3s2, false, false>(c, s, d, st->p[5], st->ga[6], G + L.iw[5])));
            RUN(chan_sum(c, s, st->ga[6], G + L.ib[5], B, 128, 9));
            int r = l6_dense_dgrad(c, s, B);
            if (r == 1) r = conv_dgrad<G3s2, false>(c, s, d, st->ga[6], P + L.iw[5], st->gp[5]);
            RUN(r);
        }
        bool ga1_summed = false;
        for (int l = 5; l >= 2; --l) {
            const int hin = l == 2 ? hs[0] : hs[l - 2];       // side of conv l's output (= its input, stride 1)
            const long n = (long)B * kICh[l] * hin * hin;
            const int hp = hs[l - 1];
            const int pblocks = (hp * hp + 255) / 256;
            const bool fused = hin == 2 * hp && (long)B * kICh[l] * pblocks <= kBiasSlabFloats;      // even map: windows tile it exactly
            if (fused) {     // gact and the bias gradient's partial sums in one pass over the activations
                float* part = bs_take(c, s, (long)B * kICh[l] * pblocks);
                hipLaunchKernelGGL(pool_relu_bwd2_kernel, dim3(pblocks, B * kICh[l]), dim3(256), 0, s, st->a[l], st->gp[l], st->ga[l],
                                   part, hin, hp);
                IT_CHECK(c);
                RUN(slab_reduce(c, s, G + L.ib[l - 1], part, kICh[l], B, (long)kICh[l] * pblocks, pblocks));
            } else {
                hipLaunchKernelGGL(pool_relu_bwd_kernel, g1(n), dim3(256), 0, s, st->a[l], st->gp[l], st->ga[l], n, hin, hs[l - 1]);
                IT_CHECK(c);
            }
            const ConvDims d = img_dims(st, l, B);
            const float* xin = l == 2 ? st->a[1] : st->p[l - 1];
            {
                int r = st->bf16 ? img_bf16_wgrad(c, s, l, hin, xin, st->ga[l], G + L.iw[l - 1], st->slab, B) : 1;
                if (r == 1) r = conv_wgrad<G3s1, false, false>(c, s, d, xin, st->ga[l], G + L.iw[l - 1]);
                RUN(r);
            }
            if (!fused) RUN(chan_sum(c, s, st->ga[l], G + L.ib[l - 1], B, kICh[l], hin * hin));
            float* dx = l == 2 ? st->ga[1] : st->gp[l - 1];
            {
                // (layer 2's data gradient is ga[1]: its channel sums are conv 1's bias gradient)
                int nparts = 0;
                float* part = l == 2 && st->bf16 ? bs_take(c, s, kPartFloats) : nullptr;
                int r = st->bf16 ? img_bf16_conv(c, s, l, hin, 1, st->ga[l], P + L.iw[l - 1], nullptr, l == 2 ? st->a[1] : nullptr, dx,
                                                 part, &nparts, B, st->imgws, st->img_packed ? 1 : 0) : 1;
                if (r == 1) r = conv_dgrad<G3s1, false>(c, s, d, st->ga[l], P + L.iw[l - 1], dx, l == 2 ? st->a[1] : nullptr);
                RUN(r);
                if (l == 2 && nparts) {
                    if ((long)nparts * 32 > kPartFloats) { VAR_SET_ERR(c, "iTHOR backward: %d bias partials", nparts); return VAR_ERR_STATE; }
                    RUN(slab_reduce(c, s, G + L.ib[0], part, 32, nparts, 32));
                    ga1_summed = true;
                }
            }
        }
        {
            ConvDims d = img_dims(st, 1, B);
            d.xb = st->bstride;
            if (st->is_u8) {
                int r = st->bf16 ? img_bf16_wgrad1(c, s, hs[0], st->image, st->bstride, st->ga[1], G + L.iw[0], st->slab, B) : 1;
                if (r == 1) r = conv_wgrad<G3s1, true, false>(c, s, d, st->image, st->ga[1], G + L.iw[0]);
                RUN(r);
            }
            else RUN((conv_wgrad<G3s1, false, false>(c, s, d, st->image, st->ga[1], G + L.iw[0])));
            if (!ga1_summed) RUN(chan_sum(c, s, st->ga[1], G + L.ib[0], B, 32, hs[0] * hs[0]));
        }
    }
    if (nclips) {
        const float* raw = st->raw + 3 * mB;
        float* graw = st->graw + 3 * mB;
        hipLaunchKernelGGL(l2norm_bwd_kernel, g1(nclips), dim3(256), 0, s, raw, st->gemb + 3 * mB, graw, nclips);
        IT_CHECK(c);
        RUN(linear_bwd(c, s, st->hid_s2, P + L.sh_w2, graw, G + L.sh_w2, G + L.sh_b2, st->ghid_s2, nclips, 64, 3));
        RUN(relu_mask(c, s, st->ghid_s2, st->hid_s2, (long)nclips * 64));
        RUN(linear_bwd(c, s, st->hid_s1, P + L.sh_w1, st->ghid_s2, G + L.sh_w1, G + L.sh_b1, st->ghid_s1, nclips, 128, 64));
        RUN(relu_mask(c, s, st->ghid_s1, st->hid_s1, (long)nclips * 128));
        RUN(linear_bwd(c, s, st->sraw, P + L.sh_w0, st->ghid_s1, G + L.sh_w0, G + L.sh_b0, st->gsraw, nclips, kSRaw, 128));
        // GRU, backward through time
        const int rows = nclips * kSeq;
        const long dirP = L.w_ih[1] - L.w_ih[0];
        const long dirGI = (long)rows * kG3, dirH = (long)(kSeq + 1) * nclips * kGh, dirS = (long)kSeq * nclips * kGh;
        const long dirDGH = (long)kSeq * nclips * kG3;
        hipLaunchKernelGGL(gru_concat_kernel, g1((long)nclips * kSRaw), dim3(256), 0, s, st->DH, st->gsraw, nclips,
                           (long)nclips * kGh, 1);
        IT_CHECK(c);
        const int dh_split = rec_split(4 * ((nclips + 63) / 64) * 2, kG3 / GG_KC, 8);
        int whole = 0;
        if (st->bf16 && st->gru_seq) {
            // (the fp32 gate gradients have no reader in this form -- the four products take the bf16 copies, the bias sums
            //  are formed in the kernel --: 0.9 GB of stores per pass at batch 256; written only for the debug mode)
            const int r = gru_bf16_seq_bwd(c, s, st->DH, st->Hb, st->R, st->Z, st->Nn, st->GHN, st->DGI, st->DGH, nclips, 2 * st->maxB,
                                           dirGI, dirH, dirS, dirDGH, st->gruws, st->keep32 ? 1 : 0);
            if (r < 0) return r;
            whole = r == 0;
        }
        for (int step = kSeq - 1; step >= 0 && !whole; --step) {
            if (st->bf16) {      // dh = DH + dgh(step+1) W_hh, then the step's gate derivatives, in one launch
                RUN(gru_bf16_step_bwd(c, s, st->DH, st->Hb, st->R, st->Z, st->Nn, st->GHN, st->DGI, st->DGH, nclips, 2 * st->maxB, step,
                                      step == kSeq - 1 ? 0 : 1, dirGI, dirH, dirS, dirDGH, st->gruws));
                continue;
            }
            hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3((nclips * kGh + 255) / 256, 2), dim3(256), 0, s, st->DH, st->DHP,
                               step == kSeq - 1 ? 0 : dh_split,
                               st->Hb + (long)step * nclips * kGh, st->R, st->Z, st->Nn, st->GHN, st->DGI, st->DGH, nclips, step,
                               dirGI, dirH, dirS, dirDGH);
            IT_CHECK(c);
            if (step == 0) break;                               // h_0 = 0 has no consumer
            DenseP<false, true, 2> p{};
            p.M = kGh; p.N = nclips; p.K = kG3;
            p.nsplit = dh_split;
            p.A = P + L.w_hh[0]; p.sam = 1; p.sak = kGh; p.zA = dirP;
            p.Bm = st->DGH + (long)step * nclips * kG3; p.sbk = 1; p.sbn = kG3; p.zB = dirDGH;
            p.C = st->DHP; p.scm = 1; p.scn = kGh; p.zC = (long)nclips * kGh; p.sC = 2L * nclips * kGh;
            RUN(gg(c, s, p, 2));
        }
        {   // dW_hh[dir][g][j] = sum_{step,clip} DGH[dir][step,clip][g] * h_prev[dir][step,clip][j]
            DenseP<false, false, 2> p{};
            p.M = kGh; p.N = kG3; p.K = kSeq * nclips;
            p.nsplit = eff_split(p.K, p.K >= 2048 ? 8 : (p.K >= 512 ? 4 : 1));
            const long one = (long)kG3 * kGh;                       // per direction; slabs hold [split][dir][g][j]
            p.A = st->Hb; p.sam = 1; p.sak = kGh; p.zA = dirH;
            p.Bm = st->DGH; p.sbk = kG3; p.sbn = 1; p.zB = dirDGH;
            p.C = st->slab; p.scm = 1; p.scn = kGh; p.zC = one; p.sC = 2 * one;
            {
                auto q = p;
                q.zA = (long)(kSeq + 1) * nclips * kGh;
                use_bf16_copies(c, q, gru_bf16_h16(st->gruws), gru_bf16_dgh16(st->gruws, 2 * st->maxB));
                if (q.a16) p = q;
            }
            if (whole && !p.b16) { VAR_SET_ERR(c, "iTHOR backward: dW_hh needs the fp32 gate gradients the GRU pass did not write"); return VAR_ERR_STATE; }
            RUN(gg(c, s, p, 2));
            for (int d = 0; d < 2; ++d) RUN(slab_reduce(c, s, G + L.w_hh[d], st->slab + d * one, (int)one, p.nsplit, 2 * one));
            // dW_ih[dir][g][i] = sum_{clip,t} DGI[dir][clip,t][g] * X[clip,t][i]
            const long onei = (long)kG3 * kGin;
            p.M = kGin; p.N = kG3; p.K = rows;
            p.A = st->s[3]; p.sam = 1; p.sak = kGin; p.zA = 0;
            p.Bm = st->DGI; p.sbk = kG3; p.sbn = 1; p.zB = dirGI;
            p.C = st->slab; p.scm = 1; p.scn = kGin; p.zC = onei; p.sC = 2 * onei;
            p.a16 = p.b16 = 0;
            use_bf16_copies(c, p, gru_bf16_x16(st->gruws, 2 * st->maxB), gru_bf16_dgi16(st->gruws, 2 * st->maxB));
            if (whole && !p.b16) { VAR_SET_ERR(c, "iTHOR backward: dW_ih needs the fp32 gate gradients the GRU pass did not write"); return VAR_ERR_STATE; }
            RUN(gg(c, s, p, 2));
            for (int d = 0; d < 2; ++d) RUN(slab_reduce(c, s, G + L.w_ih[d], st->slab + d * onei, (int)onei, p.nsplit, 2 * onei));
        }
        for (int d = 0; d < 2; ++d) {
            if (whole) {     // the sequence kernel summed its 64-clip slices on the way: fold the slices (no second pass over DGI / DGH)
                const int ncs = (nclips + 63) / 64;
                const float* part = gru_bf16_bias_part(st->gruws, 2 * st->maxB) + (long)d * ncs * 6 * kGh;
                RUN(slab_reduce(c, s, G + L.b_ih[d], part, kG3, ncs, 6L * kGh));
                RUN(slab_reduce(c, s, G + L.b_hh[d], part + kG3, kG3, ncs, 6L * kGh));
                continue;
            }
            RUN(chan_sum(c, s, st->DGI + d * dirGI, G + L.b_ih[d], rows, kG3, 1));
            RUN(chan_sum(c, s, st->DGH + d * dirDGH, G + L.b_hh[d], rows, kG3, 1));
        }
        {   // dX[clip,t][i] = sum_dir sum_g DGI[dir][clip,t][g] * W_ih[dir][g][i]
            DenseP<false, true, 0> p{};
            p.M = kGin; p.N = rows; p.K = kG3; p.nsplit = 1;
            p.A = P + L.w_ih[0]; p.sam = 1; p.sak = kGin; p.Bm = st->DGI; p.sbk = 1; p.sbn = kG3;
            p.C = st->gs[3]; p.scm = 1; p.scn = kGin;
            const unsigned short* w16 = (const unsigned short*)gru_bf16_wih16(st->gruws, 2 * st->maxB);
            use_bf16_copies(c, p, w16, gru_bf16_dgi16(st->gruws, 2 * st->maxB));
            if (!p.a16) use_bf16_copies(c, p, nullptr, gru_bf16_dgi16(st->gruws, 2 * st->maxB));
            RUN(gg(c, s, p));
            DenseP<false, true, 1> q{};
            q.M = kGin; q.N = rows; q.K = kG3; q.nsplit = 1;
            q.A = P + L.w_ih[1]; q.sam = 1; q.sak = kGin; q.Bm = st->DGI + dirGI; q.sbk = 1; q.sbn = kG3;
            q.C = st->gs[3]; q.scm = 1; q.scn = kGin;
            use_bf16_copies(c, q, w16 + (long)kG3 * kGin, (const unsigned short*)gru_bf16_dgi16(st->gruws, 2 * st->maxB) + dirGI);
            if (!q.a16) use_bf16_copies(c, q, nullptr, (const unsigned short*)gru_bf16_dgi16(st->gruws, 2 * st->maxB) + dirGI);
            if (whole && !(p.b16 && q.b16)) { VAR_SET_ERR(c, "iTHOR backward: dX needs the fp32 gate gradients the GRU pass did not write"); return VAR_ERR_STATE; }
            RUN(gg(c, s, q));
        }
        RUN(relu_mask(c, s, st->gs[3], st->s[3], (long)rows * kGin));
        {
            const ConvDims d = snd_dims(3, nclips);
            if (!st->bf16) RUN((conv_wgrad<GS3, false, true>(c, s, d, st->s[2], st->gs[3], G + L.sw[2])));
            RUN(chan_sum(c, s, st->gs[3], G + L.sb[2], nclips * kSeq, 64, 7));
            if (st->bf16) {     // (also leaves gs[2]'s C8 image for conv 2's kernels and the channel sums of gs[2] = conv 2's bias gradient)
                int nparts = 0;
                float* part = bs_take(c, s, kPartFloats);
                RUN(snd3_bf16_dgrad(c, s, st->gs[3], P + L.sw[2], st->keep32 ? st->gs[2] : nullptr, part, &nparts, nclips, 2 * st->maxB,
                                    st->bfws));
                if ((long)nparts * 64 > kPartFloats) { VAR_SET_ERR(c, "iTHOR backward: %d bias partials", nparts); return VAR_ERR_STATE; }
                RUN(slab_reduce(c, s, G + L.sb[1], part, 64, nparts, 64));
                RUN(snd3_bf16_wgrad(c, s, G + L.sw[2], st->slab, nclips, 2 * st->maxB, st->bfws));      // (reads the dgrad's gy image)
            } else {
                RUN((conv_dgrad<GS3, true>(c, s, d, st->gs[3], P + L.sw[2], st->gs[2], st->s[2])));
            }
        }
        {
            const ConvDims d = snd_dims(2, nclips);

            if (st->bf16) {
                RUN(snd2_bf16_wgrad(c, s, G + L.sw[1], st->slab, nclips, 2 * st->maxB, st->bfws));
            } else {
                ProfScope prof(c, s, TAG_ITHOR_S2_WGRAD);
                RUN((conv_wgrad<GS2, false, false>(c, s, d, st->s[1], st->gs[2], G + L.sw[1])));
            }
            if (!st->bf16) RUN(chan_sum(c, s, st->gs[2], G + L.sb[1], nclips, 64, 150 * 13));
            if (st->bf16) {     // (its store also yields the channel sums of gs[1]: conv 1's bias gradient)
                int nparts = 0;
                float* part = bs_take(c, s, kPartFloats);
                RUN(snd2_bf16_dgrad(c, s, P + L.sw[1], st->keep32 ? st->gs[1] : nullptr, part, &nparts, nclips, 2 * st->maxB, st->bfws));
                if ((long)nparts * 64 > kPartFloats) { VAR_SET_ERR(c, "iTHOR backward: %d bias partials", nparts); return VAR_ERR_STATE; }
                RUN(slab_reduce(c, s, G + L.sb[0], part, 64, nparts, 64));
            } else {
                ProfScope prof(c, s, TAG_ITHOR_S2_DGRAD);
                RUN((conv_dgrad<GS2, false>(c, s, d, st->gs[2], P + L.sw[1], st->gs[1], st->s[1])));
            }
        }
        if (st->bf16) {     // (reads the bf16 gradient image conv 2's data gradient wrote)
            const float *pos = st->pos, *neg = st->neg;
            RUN(snd1_bf16_wgrad(c, s, pos ? pos : neg, B, pos && neg ? neg : nullptr, pos && neg ? B : 0, G + L.sw[0], st->slab,
                                2 * st->maxB, st->bfws));
        } else {
            int off = 0;
            for (int q = 0; q < 2; ++q) {
                const float* src = q == 0 ? st->pos : st->neg;
                if (!src) continue;
                RUN((conv_wgrad<GS1, false, false>(c, s, snd_dims(1, B), src, st->gs[1] + (long)off * 64 * 300 * 20, G + L.sw[0])));
                off += B;
            }
            if (!st->bf16) RUN(chan_sum(c, s, st->gs[1], G + L.sb[0], nclips, 64, 300 * 20));
        }
    }
    RUN(flush_folds(c, s));
    if (nclips && st->bf16 && st->gru_seq) RUN(gru_bf16_poison_on_timeout(c, s, G, min(L.total, 65536), 2 * st->maxB, st->gruws));
    return VAR_OK;
}

int ithor_debug_buffer(var_ctx* c, const char* name, void** ptr, long* nfloats) {
    ithor_state* st = ith(c);
    if (!st) { VAR_SET_ERR(c, "var_debug_buffer: no iTHOR plan"); return VAR_ERR_PLAN; }
    const long C2 = 2L * st->maxB;
    const long ssz[4] = {0, C2 * 64 * 300 * 20, C2 * 64 * 150 * 13, C2 * kSeq * kGin};
    for (int l = 1; l <= 3; ++l) {
        char a[8], g[8];
        snprintf(a, sizeof a, "s%d", l); snprintf(g, sizeof g, "gs%d", l);
        if (!strcmp(name, a)) { *ptr = st->s[l]; *nfloats = ssz[l]; return VAR_OK; }
        if (!strcmp(name, g)) { *ptr = st->gs[l]; *nfloats = ssz[l]; return VAR_OK; }
    }
    {   // image branch: activations a1..a6 (post-ReLU), pooled maps p2..p5, and the gradients wrt them (ga*, gp*)
        const long B = st->maxB;
        const int* hs = st->hs;
        for (int l = 1; l <= 6; ++l) {
            const long n = l <= 2 ? B * 32 * hs[0] * hs[0] : (l <= 5 ? B * kICh[l] * hs[l - 2] * hs[l - 2] : B * kIRaw);
            char a[8], g[8];
            snprintf(a, sizeof a, "a%d", l); snprintf(g, sizeof g, "ga%d", l);
            if (!strcmp(name, a)) { *ptr = st->a[l]; *nfloats = n; return VAR_OK; }
            if (!strcmp(name, g)) { *ptr = st->ga[l]; *nfloats = n; return VAR_OK; }
        }
        for (int l = 2; l <= 5; ++l) {
            const long n = B * kICh[l] * hs[l - 1] * hs[l - 1];
            char a[8], g[8];
            snprintf(a, sizeof a, "p%d", l); snprintf(g, sizeof g, "gp%d", l);
            if (!strcmp(name, a)) { *ptr = st->p[l]; *nfloats = n; return VAR_OK; }
            if (!strcmp(name, g)) { *ptr = st->gp[l]; *nfloats = n; return VAR_OK; }
        }
    }
    if (!strcmp(name, "emb")) { *ptr = st->emb; *nfloats = 9L * st->maxB; return VAR_OK; }
    if (!strcmp(name, "sraw")) { *ptr = st->sraw; *nfloats = C2 * kSRaw; return VAR_OK; }
    if (!strcmp(name, "hb")) { *ptr = st->Hb; *nfloats = 2L * (kSeq + 1) * C2 * kGh; return VAR_OK; }
    if (!strcmp(name, "gh")) { *ptr = st->GH; *nfloats = 2L * (C2 + 2048) * kG3; return VAR_OK; }
    if (!strcmp(name, "gi")) { *ptr = st->GI; *nfloats = 2L * C2 * kSeq * kG3; return VAR_OK; }
    VAR_SET_ERR(c, "var_debug_buffer: unknown iTHOR buffer '%s'", name);
    return VAR_ERR_ARG;
}

template <bool AK, bool BK>
static int debug_dense(var_ctx* c, hipStream_t s, const float* a, const float* b, float* out, int M, int N, int K, int nsplit, int add) {
    const long sam = AK ? K : 1, sak = AK ? 1 : M, sbk = BK ? 1 : N, sbn = BK ? K : 1;
    if (add & 2) {      // both operands as bf16 copies, as the schedule hands the big products over (no split)
        if (nsplit > 1 || ((long)M * K) % 8 || ((long)N * K) % 8) return VAR_ERR_ARG;
        void *a16 = nullptr, *b16 = nullptr;
        VAR_HIP_CHECK(c, hipMalloc(&a16, (size_t)M * K * 2));
        VAR_HIP_CHECK(c, hipMalloc(&b16, (size_t)N * K * 2));
        int r = gru_bf16_to_bf16(c, s, a, a16, (long)M * K);
        if (r == VAR_OK) r = gru_bf16_to_bf16(c, s, b, b16, (long)N * K);
        int took = 0;
        auto run = [&](auto p) {
            p.M = M; p.N = N; p.K = K; p.nsplit = 1;
            p.A = a; p.sam = sam; p.sak = sak; p.Bm = b; p.sbk = sbk; p.sbn = sbn; p.C = out; p.scm = 1; p.scn = M;
            use_bf16_copies(c, p, a16, b16);
            took = p.a16 && p.b16 ? (dense16::resident_a_eligible(p) ? 2 : 1) : 0;
            return gg(c, s, p);
        };
        if (r == VAR_OK) r = (add & 1) ? run(DenseP<AK, BK, 1>{}) : run(DenseP<AK, BK, 0>{});
        (void)hipStreamSynchronize(s);
        (void)hipFree(a16); (void)hipFree(b16);
        return r != VAR_OK ? r : took;
    }
    if (nsplit > 1) {
        DenseP<AK, BK, 2> p{};
        p.M = M; p.N = N; p.K = K; p.nsplit = eff_split(K, nsplit);
        p.A = a; p.sam = sam; p.sak = sak; p.Bm = b; p.sbk = sbk; p.sbn = sbn; p.C = out; p.scm = 1; p.scn = M; p.sC = (long)M * N;
        const int r = gg(c, s, p);
        return r != VAR_OK ? r : (dense16::eligible(p) ? 1 : 0);
    }
    if (add) {
        DenseP<AK, BK, 1> p{};
        p.M = M; p.N = N; p.K = K; p.nsplit = 1;
        p.A = a; p.sam = sam; p.sak = sak; p.Bm = b; p.sbk = sbk; p.sbn = sbn; p.C = out; p.scm = 1; p.scn = M;
        const int r = gg(c, s, p);
        return r != VAR_OK ? r : (dense16::eligible(p) ? 1 : 0);
    }
    DenseP<AK, BK, 0> p{};
    p.M = M; p.N = N; p.K = K; p.nsplit = 1;
    p.A = a; p.sam = sam; p.sak = sak; p.Bm = b; p.sbk = sbk; p.sbn = sbn; p.C = out; p.scm = 1; p.scn = M;
    const int r = gg(c, s, p);
    return r != VAR_OK ? r : (dense16::eligible(p) ? 1 : 0);
}

// ---- C ABI ------------------------------------------------------------------------------------------------------
#define CHECK_CTX(c) do { if (!(c)) return VAR_ERR_ARG; } while (0)

extern "C" {

int var_ithor_param_count(void) { return make_ithor_layout().total; }

int var_ithor_plan(var_ctx* c, int max_batch, int img_hw) {
    CHECK_CTX(c);
    if (max_batch < 1 || max_batch > 1024) { VAR_SET_ERR(c, "var_ithor_plan: batch %d outside 1..1024", max_batch); return VAR_ERR_ARG; }
    int hs[6];
    hs[0] = img_hw;
    for (int i = 1; i <= 4; ++i) hs[i] = hs[i - 1] / 2;
    hs[5] = (hs[4] - 1) / 2 + 1;
    if (img_hw < 16 || hs[5] != 3) {
        VAR_SET_ERR(c, "var_ithor_plan: img side %d does not end in a 3x3 map (Linear(1152, .))", img_hw);
        return VAR_ERR_ARG;
    }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    ithor_state* st = ith(c);
    if (st && st->maxB >= max_batch && st->H == img_hw) return VAR_OK;
    if (st) {      // retire (do not free) the superseded workspace: captured graphs may still replay on it
        if (st->ws) { int rc = retire_block(c, st->ws); if (rc != VAR_OK) return rc; }
        delete st;
        c->ith = nullptr;
    }
    c->plan_gen++;
    st = new ithor_state();
    c->ith = st;
    st->L = make_ithor_layout();
    st->maxB = max_batch; st->H = img_hw;
    memcpy(st->hs, hs, sizeof(hs));
    const long B = max_batch, C2 = 2 * B;
    // sizes (floats)
    long asz[7] = {0}, psz[6] = {0};
    asz[1] = B * 32 * hs[0] * hs[0]; asz[2] = asz[1];
    for (int l = 3; l <= 5; ++l) asz[l] = B * kICh[l] * hs[l - 2] * hs[l - 2];
    asz[6] = B * kIRaw;
    for (int l = 2; l <= 5; ++l) psz[l] = B * kICh[l] * hs[l - 1] * hs[l - 1];
    const long ssz[4] = {0, C2 * 64 * 300 * 20, C2 * 64 * 150 * 13, C2 * kSeq * kGin};
    const long rows = C2 * kSeq;
    long total = 0;
    auto take = [&](long n) { long o = total; total += (n + 63) & ~63L; return o; };
    long oa[7], oga[7], op[6], ogp[6], os[4], ogs[4];
    for (int l = 1; l <= 6; ++l) { oa[l] = take(asz[l]); oga[l] = take(asz[l]); }
    for (int l = 2; l <= 5; ++l) { op[l] = take(psz[l]); ogp[l] = take(psz[l]); }
    for (int l = 1; l <= 3; ++l) { os[l] = take(ssz[l]); ogs[l] = take(ssz[l]); }
    const long oGI = take(2 * rows * kG3), oDGI = take(2 * rows * kG3), oDGH = take(2 * rows * kG3);
    const long oGH = take(2 * (C2 + 2048) * kG3), oHb = take(2 * (kSeq + 1) * C2 * kGh);    // GH: up to 8 split-K slabs
    const long oDHP = take(2 * (C2 + 6144) * kGh), oslab = take(kSlabFloats), obslab = take(kBiasSlabFloats);
    const long oR = take(2 * rows * kGh), oZ = take(2 * rows * kGh), oN = take(2 * rows * kGh), oGHN = take(2 * rows * kGh);
    const long oDH = take(2 * C2 * kGh), osraw = take(C2 * kSRaw), ogsraw = take(C2 * kSRaw);
    const long ohi = take(B * 128), oghi = take(B * 128), ohs1 = take(C2 * 128), oghs1 = take(C2 * 128);
    const long ohs2 = take(C2 * 64), oghs2 = take(C2 * 64);
    const long oraw = take(9 * B), ograw = take(9 * B), oemb = take(9 * B), ogemb = take(9 * B), oloss = take(64);
    const long obf = take((snd_bf16_workspace_bytes((int)C2) + 3) / 4), ogru = take((gru_bf16_workspace_bytes((int)C2) + 3) / 4);
    const long oimg = take((img_bf16_workspace_bytes() + 3) / 4);
    const long kx6 = 128L * 36;                                // layer 6 as a dense layer: room for the 6 x 6 map
    const long ol6w = take(kIRaw * kx6 / 2), ol6x = take(B * kx6 / 2), ol6g = take(B * kIRaw / 2 + 8), ol6b = take(kIRaw);
    VAR_HIP_CHECK(c, hipMalloc((void**)&st->ws, (size_t)total * sizeof(float)));
    float* w = (float*)st->ws;
    for (int l = 1; l <= 6; ++l) { st->a[l] = w + oa[l]; st->ga[l] = w + oga[l]; }
    for (int l = 2; l <= 5; ++l) { st->p[l] = w + op[l]; st->gp[l] = w + ogp[l]; }
    for (int l = 1; l <= 3; ++l) { st->s[l] = w + os[l]; st->gs[l] = w + ogs[l]; }
    st->GI = w + oGI; st->DGI = w + oDGI; st->DGH = w + oDGH; st->GH = w + oGH; st->Hb = w + oHb;
    st->DHP = w + oDHP; st->slab = w + oslab; st->bslab = w + obslab;
    st->R = w + oR; st->Z = w + oZ; st->Nn = w + oN; st->GHN = w + oGHN; st->DH = w + oDH;
    st->sraw = w + osraw; st->gsraw = w + ogsraw;
    st->hid_i = w + ohi; st->ghid_i = w + oghi; st->hid_s1 = w + ohs1; st->ghid_s1 = w + oghs1;
    st->hid_s2 = w + ohs2; st->ghid_s2 = w + oghs2;
    st->bfws = w + obf; st->gruws = w + ogru; st->imgws = w + oimg;
    st->l6w = (unsigned short*)(w + ol6w); st->l6x = (unsigned short*)(w + ol6x); st->l6g = (unsigned short*)(w + ol6g); st->l6b = w + ol6b;
    st->raw = w + oraw; st->graw = w + ograw; st->emb = w + oemb; st->gemb = w + ogemb; st->loss = w + oloss;
    RUN(gru_bf16_reset_timeout(c, nullptr, (int)C2, st->gruws));
    VAR_HIP_CHECK(c, hipStreamSynchronize(nullptr));
    ithor_update_guard(c);
    return VAR_OK;
}

static int ithor_check(var_ctx* c, int B, int H, const char* who) {
    ithor_state* st = ith(c);
    if (!st || B > st->maxB || H != st->H) { VAR_SET_ERR(c, "%s: var_ithor_plan(%d, %d) first", who, B, H); return VAR_ERR_PLAN; }
    if (B < 1) { VAR_SET_ERR(c, "%s: empty batch", who); return VAR_ERR_ARG; }
    return VAR_OK;
}

// Adam over this model's arena skips its update while the current step's time-out word is set (pack_adam.hip)
static void ithor_update_guard(var_ctx* c) {
    ithor_state* st = ith(c);
    const bool on = st && st->bf16 && st->gru_seq && st->gruws;
    c->adam_guard = on ? gru_bf16_timeout_ptr(2 * st->maxB, st->gruws) : nullptr;
    c->adam_guard_n = st ? (long)st->L.total : 0;          // (also what the loss guard, var_ithor_guard_loss, applies to)
}

static int copy_out(var_ctx* c, hipStream_t s, const float* src, float* dst, long n) {
    if (!dst || n <= 0) return VAR_OK;
    hipLaunchKernelGGL(copy_rows_kernel, g1(n), dim3(256), 0, s, src, dst, n);
    IT_CHECK(c);
    return VAR_OK;
}

int var_ithor_set_bf16(var_ctx* c, int on) {
    CHECK_CTX(c);
    ithor_state* st = ith(c);
    if (!st) { VAR_SET_ERR(c, "var_ithor_set_bf16: var_ithor_plan first"); return VAR_ERR_PLAN; }
    const int old = st->bf16 ? (st->keep32 ? 2 : 1) : 0;
    if (on >= 0) { st->bf16 = on != 0; st->keep32 = on == 2; ithor_update_guard(c); }
    return old;
}

int var_ithor_set_gru_sequence(var_ctx* c, int on) {
    CHECK_CTX(c);
    ithor_state* st = ith(c);
    if (!st) { VAR_SET_ERR(c, "var_ithor_set_gru_sequence: var_ithor_plan first"); return VAR_ERR_PLAN; }
    const int old = st->gru_seq ? 1 : 0;
    if (on >= 0) {
        st->gru_seq = on != 0;
        VAR_HIP_CHECK(c, hipSetDevice(c->device));
        RUN(gru_bf16_reset_timeout(c, nullptr, 2 * st->maxB, st->gruws));      // (setting the form also clears the status word)
        VAR_HIP_CHECK(c, hipStreamSynchronize(nullptr));
        ithor_update_guard(c);
    }
    return old;
}

int var_ithor_guard_loss(var_ctx* c, const float* loss_dev) {
    CHECK_CTX(c);
    ithor_state* st = ith(c);
    if (!st) { VAR_SET_ERR(c, "var_ithor_guard_loss: var_ithor_plan first"); return VAR_ERR_PLAN; }
    c->adam_guard_loss = loss_dev;
    return VAR_OK;
}

int var_debug_ithor_gru_drop_workgroup(var_ctx* c) {
    CHECK_CTX(c);
    ithor_state* st = ith(c);
    if (!st) { VAR_SET_ERR(c, "var_debug_ithor_gru_drop_workgroup: var_ithor_plan first"); return VAR_ERR_PLAN; }
    st->gru_drop_one = true;
    return VAR_OK;
}

int var_ithor_gru_status(var_ctx* c, unsigned* word) {
    CHECK_CTX(c);
    ithor_state* st = ith(c);
    if (!st || !word) { VAR_SET_ERR(c, "var_ithor_gru_status: var_ithor_plan first"); return VAR_ERR_PLAN; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    return gru_bf16_timeout_word(c, 2 * st->maxB, st->gruws, word);
}

int var_ithor_encoder_fwd(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8,
                          long image_bstride, const float* snd_pos, const float* snd_neg, int B, int H,
                          float* image_feat, float* pos_feat, float* neg_feat, float* image_raw, float* pos_raw,
                          int save_for_bwd) {
    CHECK_CTX(c);
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    if (!params) { VAR_SET_ERR(c, "var_ithor_encoder_fwd: params is NULL"); return VAR_ERR_ARG; }
    RUN(ithor_check(c, B, H, "var_ithor_encoder_fwd"));
    hipStream_t s = (hipStream_t)stream;
    ithor_state* st = ith(c);
    RUN(ithor_fwd(c, s, params, image, image_is_u8, image_bstride, snd_pos, snd_neg, B, save_for_bwd != 0));
    const long mB = st->maxB;
    if (image) {
        RUN(copy_out(c, s, st->emb, image_feat, 3L * B));
        RUN(copy_out(c, s, st->a[6], image_raw, (long)kIRaw * B));
    }
    int off = 0;
    if (snd_pos) {
        RUN(copy_out(c, s, st->emb + 3 * mB, pos_feat, 3L * B));
        RUN(copy_out(c, s, st->sraw, pos_raw, (long)kSRaw * B));
        off = B;
    }
    if (snd_neg) RUN(copy_out(c, s, st->emb + 3 * mB + 3L * off, neg_feat, 3L * B));
    if (!save_for_bwd) st->B = 0;              // the recurrent states were not kept: no backward from this forward
    return VAR_OK;
}

int var_debug_ithor_dense(var_ctx* c, void* stream, int a_kfast, int b_kfast, const float* a, const float* b, float* out, int M,
                          int N, int K, int nsplit, int add) {
    CHECK_CTX(c);
    if (!ith(c) || !ith(c)->bf16) { VAR_SET_ERR(c, "var_debug_ithor_dense: var_ithor_plan + var_ithor_set_bf16(1) first"); return VAR_ERR_PLAN; }
    if (!a || !b || !out || M < 1 || N < 1 || K < 1) return VAR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (a_kfast) return b_kfast ? debug_dense<true, true>(c, s, a, b, out, M, N, K, nsplit, add)
                                : debug_dense<true, false>(c, s, a, b, out, M, N, K, nsplit, add);
    return b_kfast ? debug_dense<false, true>(c, s, a, b, out, M, N, K, nsplit, add)
                   : debug_dense<false, false>(c, s, a, b, out, M, N, K, nsplit, add);
}

int var_ithor_saved_generation(var_ctx* c) {
    if (!c) return VAR_ERR_ARG;
    ithor_state* st = ith(c);
    return (st && st->B > 0) ? st->gen : 0;
}

int var_ithor_encoder_bwd(var_ctx* c, void* stream, const float* params, const float* g_image_feat,
                          const float* g_pos_feat, const float* g_neg_feat, float* grads) {
    CHECK_CTX(c);
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    ithor_state* st = ith(c);
    if (!st || st->B == 0) { VAR_SET_ERR(c, "var_ithor_encoder_bwd: no saved forward"); return VAR_ERR_STATE; }
    if (!params || !grads) { VAR_SET_ERR(c, "var_ithor_encoder_bwd: NULL argument"); return VAR_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const long mB = st->maxB; const int B = st->B;
    RUN(var_zero_async(c, s, st->gemb, sizeof(float) * 9 * mB));
    if (st->has_img && g_image_feat) RUN(copy_out(c, s, g_image_feat, st->gemb, 3L * B));
    int off = 0;
    if (st->has_pos) { if (g_pos_feat) RUN(copy_out(c, s, g_pos_feat, st->gemb + 3 * mB, 3L * B)); off = B; }
    if (st->has_neg && g_neg_feat) RUN(copy_out(c, s, g_neg_feat, st->gemb + 3 * mB + 3L * off, 3L * B));
    return ithor_bwd(c, s, params, grads);
}

int var_ithor_loss_grad(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8,
                        long image_bstride, const float* snd_pos, const float* snd_neg, int B, int H, float margin,
                        float inv_count, float* grads, float* loss_out, float* feats_out) {
    CHECK_CTX(c);
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    if (!params || !image || !snd_pos || !snd_neg || !grads) {
        VAR_SET_ERR(c, "var_ithor_loss_grad: params, image, both sounds and grads are required");
        return VAR_ERR_ARG;
    }
    RUN(ithor_check(c, B, H, "var_ithor_loss_grad"));
    hipStream_t s = (hipStream_t)stream;
    ithor_state* st = ith(c);
    const long mB = st->maxB;
    RUN(ithor_fwd(c, s, params, image, image_is_u8, image_bstride, snd_pos, snd_neg, B, true));
    float* lo = loss_out ? loss_out : st->loss;
    RUN(launch_triplet(c, s, st->emb, st->emb + 3 * mB, st->emb + 3 * mB + 3L * B, B, margin, inv_count, lo, st->gemb,
                       st->gemb + 3 * mB, st->gemb + 3 * mB + 3L * B));
    // (the hinge swallows NaN embeddings -- max(0, NaN) = 0 --: a forward whose GRU hand-off expired must not report loss 0)
    if (st->bf16 && st->gru_seq) RUN(gru_bf16_poison_on_timeout(c, s, lo, 1, 2 * st->maxB, st->gruws));
    if (feats_out) {
        // (B,9) = [a | p | n]
        VAR_HIP_CHECK(c, hipMemcpy2DAsync(feats_out, 9 * sizeof(float), st->emb, 3 * sizeof(float), 3 * sizeof(float), B,
                                          hipMemcpyDeviceToDevice, s));
        VAR_HIP_CHECK(c, hipMemcpy2DAsync(feats_out + 3, 9 * sizeof(float), st->emb + 3 * mB, 3 * sizeof(float),
                                          3 * sizeof(float), B, hipMemcpyDeviceToDevice, s));
        VAR_HIP_CHECK(c, hipMemcpy2DAsync(feats_out + 6, 9 * sizeof(float), st->emb + 3 * mB + 3L * B, 3 * sizeof(float),
                                          3 * sizeof(float), B, hipMemcpyDeviceToDevice, s));
    }
    return ithor_bwd(c, s, params, grads);
}

}  // extern "C"
