// Embedding heads, L2 normalisation and the triplet loss, forward and backward:
//   imgTriplet / soundTriplet = Linear(K,128)+ReLU+Linear(128,3)   (arm_pretext_model.py:46-56)
//   F.normalize(p=2, dim=1, eps=1e-12)                              (pretext_base.py:18,23)
//   TripletMarginLoss(margin, p=2, eps=1e-6, mean)                  (VAR/pretext_VAR.py:38,64)
//
// The 128-wide hidden layer is GEMM-shaped and runs on the f32 matrix cores
// (v_mfma_f32_32x32x2_f32); the 3-wide output layer, the normalisation and the loss are
// per-row VALU work.  Every MFMA operand is read either from LDS (padded, conflict-free) or as
// coalesced 128-byte rows straight from L2, prefetched a block of k-steps ahead in registers.
// All reductions have a fixed order (bitwise reproducible).
#include "var_common.h"

namespace {

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

// ------------------------------------------------------------------------------------------
// backward, elementwise part: one thread per (row, hidden unit)
//   graw = normalise-bwd(gemb);  ghid = (graw @ W1) * (hid > 0)  in both [row][n] and [n][row]
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128)
heads_bwd_rows_kernel(int R, const float* __restrict__ w1, const float* __restrict__ hid,
                      const float* __restrict__ emb_raw, const float* __restrict__ emb,
                      const float* __restrict__ gemb, float* __restrict__ graw_out,
                      float* __restrict__ ghid, float* __restrict__ ghidT) {
    const int row = blockIdx.x, n = threadIdx.x;
    const float a = emb_raw[row * 3], b = emb_raw[row * 3 + 1], c = emb_raw[row * 3 + 2];
    const float nrm = sqrtf(a * a + b * b + c * c);
    const float den = nrm > 1e-12f ? nrm : 1e-12f;
    const float y0 = emb[row * 3], y1 = emb[row * 3 + 1], y2 = emb[row * 3 + 2];
    const float e0 = gemb[row * 3], e1 = gemb[row * 3 + 1], e2 = gemb[row * 3 + 2];
    const float dot = y0 * e0 + y1 * e1 + y2 * e2;
    const float g0 = (e0 - y0 * dot) / den, g1 = (e1 - y1 * dot) / den, g2 = (e2 - y2 * dot) / den;
    if (n < 4) graw_out[row * 4 + n] = n == 0 ? g0 : (n == 1 ? g1 : (n == 2 ? g2 : 0.f));
    float v = g0 * w1[n] + g1 * w1[kHid + n] + g2 * w1[2 * kHid + n];
    if (!(hid[(size_t)row * kHid + n] > 0.f)) v = 0.f;
    ghid[(size_t)row * kHid + n] = v;
    ghidT[(size_t)n * R + row] = v;
}

// ------------------------------------------------------------------------------------------
// The training step's form of the two kernels above plus heads_finish_kernel and triplet_kernel: a row's triplet
// gradient depends only on its own sample (a_i, p_i, n_i), so every row block finishes the three embeddings of its
// sample from the forward partials, forms the loss gradient of its own role (image / positive / negative) and
// carries on with the row-wise backward.  The critical path loses two launches and a cross-stream hand-over:
// the loss VALUE (triplet_loss_kernel) is computed beside it on the side stream.
// ------------------------------------------------------------------------------------------
struct Emb3 { float raw[3], y[3], den; };
__device__ __forceinline__ Emb3 finish_emb(const float* part_row, const float* __restrict__ b1) {
    const float4* p = (const float4*)part_row;
    const float4 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
    Emb3 e;
    e.raw[0] = ((p0.x + p1.x) + (p2.x + p3.x)) + b1[0];
    e.raw[1] = ((p0.y + p1.y) + (p2.y + p3.y)) + b1[1];
    e.raw[2] = ((p0.z + p1.z) + (p2.z + p3.z)) + b1[2];
    const float nrm = sqrtf(e.raw[0] * e.raw[0] + e.raw[1] * e.raw[1] + e.raw[2] * e.raw[2]);
    e.den = nrm > 1e-12f ? nrm : 1e-12f;
#pragma unroll
    for (int d = 0; d < 3; ++d) e.y[d] = e.raw[d] / e.den;
    return e;
}
// loss term of one sample and the gradients wrt the three normalised embeddings (triplet_kernel's arithmetic)
__device__ __forceinline__ float triplet_row(const Emb3& a, const Emb3& p, const Emb3& n, float margin, float inv_count,
                                             float (&ga)[3], float (&gp)[3], float (&gn)[3]) {
    float dp[3], dn[3], sp = 0.f, sn = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        dp[d] = (a.y[d] - p.y[d]) + 1e-6f;
        dn[d] = (a.y[d] - n.y[d]) + 1e-6f;
        sp += dp[d] * dp[d];
        sn += dn[d] * dn[d];
    }
    const float dap = sqrtf(sp), dan = sqrtf(sn);
    const float l = dap - dan + margin;
    const bool active = l > 0.f;
    const float s = active ? inv_count : 0.f;
    const float ip = dap > 0.f ? 1.f / dap : 0.f, in_ = dan > 0.f ? 1.f / dan : 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float up = dp[d] * ip, un = dn[d] * in_;
        ga[d] = s * (up - un);
        gp[d] = -s * up;
        gn[d] = s * un;
    }
    return active ? l : 0.f;
}

// ------------------------------------------------------------------------------------------
// Device-side hand-over between the caller's stream and the side stream (var_ctx::jsig; join_* in var_common.h).  In a replayed
// step the first backward kernel on the caller's stream -- the image rows below -- needs the sound embeddings' partials from the
// side stream, and the sound rows on the side stream need the image's from the caller's (img_mid3.hip leaves them).
// As a graph edge such a dependency is a barrier packet in front of the kernel, and the queue takes ~10 us over it even when the
// other side finished long before (the timeline: conv 3-5 ends at 114.6 us, sound heads at 110.0, rows start at 125.3); a kernel
// with a successor on the other queue costs its same-queue successor ~6 us as well.  So the training step leaves both edges
// out -- between its fork and its final join no graph edge crosses the streams: the last workgroup of the sound heads' forward
// counts sig[1] up; the conv 3-5 kernel's last workgroup, on the caller's stream, does not end before it has seen that (it has
// happened by then unless the sound branch is late); the image rows read the partials with agent-scope loads.  The other
// direction is the same with sig[4..7]: counted up by the last workgroup of the conv 3-5 kernel, awaited by the sound heads'
// last workgroup.  A wait gives up after 5 ms and counts itself in sig[3] / sig[7] (var_join_status): the step's numbers are
// then undefined, nothing hangs.
// ------------------------------------------------------------------------------------------
// Who polls: the LAST workgroup of the kernel in front of the consumer on its own stream (var_common.h: join_signal) -- the conv
// 3-5 kernel waits for the sound heads' flag before it ends, the sound heads for the conv 3-5 kernel's; the rows kernels behind
// them start with everything in place.  (Polling in the rows kernels themselves -- one workgroup per row -- hung a step whose
// sound branch had run ahead until the time-out: 512 resident workgroups, a few VGPRs and 512 B of LDS each, sat on every CU,
// while the conv 3-5 kernel they were waiting for needs a CU's WHOLE LDS and register file per workgroup and could not be placed
// anywhere.  A one-wave gate kernel in front of the rows is safe but costs what the edge did, ~8 us per kernel boundary at that
// point of the step; sixteen persistent polling workgroups walking the rows are safe and take 16 us instead of 6.)

// rows [g0, g0 + gridDim.x) of the (3B) stack [image | positive | negative]; part = (3B,4,4) forward partials
// sig: the partials of the rows in sig_rows (bit r: image / positive / negative) come from the other stream (complete: see
// above) and are read with agent-scope loads; nullptr: one stream.  loss_terms: sample i's loss term goes to loss_terms[i]
__global__ void __launch_bounds__(128)
heads_bwd_rows_fused_kernel(int R, int B, int g0, const float* __restrict__ w1, const float* __restrict__ hid,
                            const float* part, const float* __restrict__ b1_img,
                            const float* __restrict__ b1_snd, float margin, float inv_count,
                            float* __restrict__ emb_raw, float* __restrict__ emb, float* __restrict__ graw_out,
                            float* __restrict__ ghid, float* __restrict__ ghidT, unsigned* sig, int sig_rows,
                            float* __restrict__ loss_terms) {
    const int row = blockIdx.x, n = threadIdx.x;
    const int g = g0 + row, role = g / B, i = g - role * B;
    __shared__ __attribute__((aligned(16))) float sp[48];      // the sample's three partial rows
    if (n < 48) {
        const int r = n >> 4;
        const float* src = part + (size_t)(r * B + i) * 16 + (n & 15);
        sp[n] = (sig && ((sig_rows >> r) & 1)) ? join_load(src) : *src;
    }
    __syncthreads();
    const Emb3 ea = finish_emb(sp, b1_img);
    const Emb3 ep = finish_emb(sp + 16, b1_snd);
    const Emb3 en = finish_emb(sp + 32, b1_snd);
    float ga[3], gp[3], gn[3];
    const float lterm = triplet_row(ea, ep, en, margin, inv_count, ga, gp, gn);
    if (loss_terms && n == 0) loss_terms[i] = lterm;           // (the image rows: summed by heads_bwd_gemm_kernel's last block)
    const Emb3& me = role == 0 ? ea : (role == 1 ? ep : en);
    const float e0 = role == 0 ? ga[0] : (role == 1 ? gp[0] : gn[0]);
    const float e1 = role == 0 ? ga[1] : (role == 1 ? gp[1] : gn[1]);
    const float e2 = role == 0 ? ga[2] : (role == 1 ? gp[2] : gn[2]);
    if (n < 3) { emb_raw[g * 3 + n] = me.raw[n]; emb[g * 3 + n] = me.y[n]; }
    const float dot = me.y[0] * e0 + me.y[1] * e1 + me.y[2] * e2;
    const float q0 = (e0 - me.y[0] * dot) / me.den, q1 = (e1 - me.y[1] * dot) / me.den, q2 = (e2 - me.y[2] * dot) / me.den;
    if (n < 4) graw_out[row * 4 + n] = n == 0 ? q0 : (n == 1 ? q1 : (n == 2 ? q2 : 0.f));
    float v = q0 * w1[n] + q1 * w1[kHid + n] + q2 * w1[2 * kHid + n];
    if (!(hid[(size_t)row * kHid + n] > 0.f)) v = 0.f;
    ghid[(size_t)row * kHid + n] = v;
    ghidT[(size_t)n * R + row] = v;
}

// loss_out[0] = inv_count * sum_i max(||a-p+eps|| - ||a-n+eps|| + margin, 0) from the forward partials (fixed order)
__global__ void __launch_bounds__(256)
triplet_loss_kernel(const float* __restrict__ part, const float* __restrict__ b1_img, const float* __restrict__ b1_snd,
                    int B, float margin, float inv_count, float* __restrict__ loss_out) {
    __shared__ float psum[4];
    const int tid = threadIdx.x;
    float local = 0.f;
    for (int i = tid; i < B; i += 256) {
        const Emb3 ea = finish_emb(part + (size_t)i * 16, b1_img);
        const Emb3 ep = finish_emb(part + (size_t)(B + i) * 16, b1_snd);
        const Emb3 en = finish_emb(part + (size_t)(2 * B + i) * 16, b1_snd);
        float ga[3], gp[3], gn[3];
        local += triplet_row(ea, ep, en, margin, inv_count, ga, gp, gn);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((tid & 63) == 0) psum[tid >> 6] = local;
    __syncthreads();
    if (tid == 0) loss_out[0] = ((psum[0] + psum[1]) + (psum[2] + psum[3])) * inv_count;
}

// ------------------------------------------------------------------------------------------
// backward, GEMM part: one wave per 32x32 output tile, D[i][j] = sum_q A[q][i] * Bm[q][j]
// with BOTH operands stored [q][free] (free index contiguous => 128-byte coalesced rows):
//   dW0[n][k]  : A = ghid [row][n],  Bm = x   [row][k],  q = row   (+ db0 = column sums of A)
//   dW1[d][n]  : A = graw [row][4],  Bm = hid [row][n],  q = row   (+ db1)
//   gx [row][k]: A = ghidT[n][row],  Bm = W0  [n][k],    q = n,    masked by x > 0
// ------------------------------------------------------------------------------------------
struct TileOp {
    const float* A; int lda; int alim;       // alim: valid extent of A's free index (from its tile origin)
    const float* Bm; int ldb;
    int Q;                                   // reduction length
};

// this wave's share of the reduction: steps s = wave, wave+4, ... (two q per step).  Only A is predicated
// (a zero A kills the product); B rows are merely clamped to stay in bounds.
__device__ __forceinline__ void tile_gemm(const TileOp& t, int lane, int wave, f32x16& acc, float& asum) {
    constexpr int UB = 8;
    const int half = lane >> 5, l31 = lane & 31;
    const bool aok = l31 < t.alim;
    const float* ap = t.A + (aok ? l31 : 0);
    const float* bp = t.Bm + l31;
    const int steps = (t.Q + 1) / 2;
    const int mine = (steps - wave + 3) / 4;            // steps of this wave
    float ab[2][UB], bb[2][UB];
    auto fetch = [&](int buf, int i0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int q = 2 * (wave + 4 * (i0 + u)) + half;
            const int qc = q < t.Q ? q : t.Q - 1;
            ab[buf][u] = ap[(size_t)qc * t.lda];
            bb[buf][u] = bp[(size_t)qc * t.ldb];
        }
    };
    auto mul = [&](int buf, int i0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int q = 2 * (wave + 4 * (i0 + u)) + half;
            const float a = (aok && q < t.Q && i0 + u < mine) ? ab[buf][u] : 0.f;
            asum += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[buf][u], acc, 0, 0, 0);
        }
    };
    fetch(0, 0);
#pragma unroll 1
    for (int i0 = 0; i0 < mine; i0 += 2 * UB) {
        fetch(1, i0 + UB);
        __builtin_amdgcn_sched_barrier(0);
        mul(0, i0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, i0 + 2 * UB);
        __builtin_amdgcn_sched_barrier(0);
        mul(1, i0 + UB);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K>
__global__ void __launch_bounds__(256)
heads_bwd_gemm_kernel(int R /*rows in this call*/, int RT /*row stride of ghidT*/, const float* __restrict__ x,
                      const float* __restrict__ w0, const float* __restrict__ hid,
                      const float* __restrict__ graw, const float* __restrict__ ghid,
                      const float* __restrict__ ghidT, float* __restrict__ dw0, float* __restrict__ db0,
                      float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ gx,
                      const float* __restrict__ loss_terms, float inv_count, float* __restrict__ loss_out) {
    constexpr int KB = K / 32;
    __shared__ float red[4][1024];
    __shared__ float reda[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x;                         // one tile per workgroup, 4 waves split its reduction
    const int RB = (R + 31) / 32;
    const int n_dw0 = 4 * KB, n_dw1 = 4;
    if (loss_terms && tile == (int)gridDim.x - 1) {
        // one more workgroup: the loss VALUE, loss_out[0] = inv_count * sum_i term_i in triplet_loss_kernel's fixed order (the
        // terms come from the rows kernel in front of this one; as a kernel of its own on the side stream the sum stood between
        // the sound heads and the sound CNN's backward, 10-25 us of that stream's critical path)
        float local = 0.f;
        for (int i = tid; i < R; i += 256) local += loss_terms[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
        if (lane == 0) reda[0][wave] = local;
        __syncthreads();
        if (tid == 0) loss_out[0] = ((reda[0][0] + reda[0][1]) + (reda[0][2] + reda[0][3])) * inv_count;
        return;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    int kind, i0 = 0, i1 = 0;                            // tile coordinates
    if (tile < n_dw0) {
        kind = 0; i0 = tile / KB; i1 = tile - i0 * KB;   // nb, kb
        TileOp t{ghid + i0 * 32, kHid, 32, x + i1 * 32, K, R};
        tile_gemm(t, lane, wave, acc, asum);
    } else if (tile < n_dw0 + n_dw1) {
        kind = 1; i0 = tile - n_dw0;                     // nb
        TileOp t{graw, 4, 3, hid + i0 * 32, kHid, R};
        tile_gemm(t, lane, wave, acc, asum);
    } else {
        kind = 2;
        const int g = tile - n_dw0 - n_dw1;
        i0 = g / KB; i1 = g - i0 * KB;                   // rb, kb
        TileOp t{ghidT + i0 * 32, RT, R - i0 * 32, w0 + i1 * 32, K, kHid};
        tile_gemm(t, lane, wave, acc, asum);
    }
    (void)RB;
    // fold the four K slices (fixed order)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
    asum += __shfl_down(asum, 32, 64);
    if (half == 0) reda[wave][l31] = asum;
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        const int i = e >> 5, j = e & 31;                // D[i][j]
        if (kind == 0) {
            dw0[(size_t)(i0 * 32 + i) * K + i1 * 32 + j] = v;
        } else if (kind == 1) {
            if (i < 3) dw1[i * kHid + i0 * 32 + j] = v;
        } else {
            const int row = i0 * 32 + i;
            if (row < R) {
                const size_t o = (size_t)row * K + i1 * 32 + j;
                gx[o] = x[o] > 0.f ? v : 0.f;
            }
        }
    }
    if (tid < 32) {
        const float v = (reda[0][tid] + reda[1][tid]) + (reda[2][tid] + reda[3][tid]);
        if (kind == 0 && i1 == 0) db0[i0 * 32 + tid] = v;
        if (kind == 1 && i0 == 0 && tid < 3) db1[tid] = v;
    }
}

// ------------------------------------------------------------------------------------------
// The training step's backward of one head in ONE launch: heads_bwd_rows_fused_kernel's row work inside heads_bwd_gemm_kernel's
// workgroups.  Both kernels are at the floor of what a launch costs at this point of the step (6.6 + 8.1 us with the chip nearly
// idle, the data-gradient chain waiting behind them), and the rows are cheap: every workgroup works out the three scalars
// (the gradient wrt the raw embedding) of the rows ITS tile reduces over -- all R for a weight-gradient tile, one per thread; its
// own 32 for a data-gradient tile -- and forms its A operand from them on the fly:
//   ghid[row][n] = (q0 w1[0][n] + q1 w1[1][n] + q2 w1[2][n]) . (hid[row][n] > 0)         (the rows kernel's expression, same order)
// so ghid / ghidT / graw never exist in memory.  The products, their order and the folds are heads_bwd_gemm_kernel's.
// The workgroup of the (dW1, block 0) tile also leaves the rows' embeddings (emb_raw, emb); one more workgroup sums the loss.
// ------------------------------------------------------------------------------------------
constexpr int kFusedMaxRows = 1024;
struct RowQ { float q0, q1, q2, lterm; };
__device__ __forceinline__ RowQ row_scalars(int g, int B, const float* part, const float* __restrict__ b1_img, const float* __restrict__ b1_snd,
                                            float margin, float inv_count, Emb3& me) {
    const int role = g / B, i = g - role * B;
    const Emb3 ea = finish_emb(part + (size_t)i * 16, b1_img);
    const Emb3 ep = finish_emb(part + (size_t)(B + i) * 16, b1_snd);
    const Emb3 en = finish_emb(part + (size_t)(2 * B + i) * 16, b1_snd);
    float ga[3], gp[3], gn[3];
    RowQ r;
    r.lterm = triplet_row(ea, ep, en, margin, inv_count, ga, gp, gn);
    me = role == 0 ? ea : (role == 1 ? ep : en);
    const float e0 = role == 0 ? ga[0] : (role == 1 ? gp[0] : gn[0]);
    const float e1 = role == 0 ? ga[1] : (role == 1 ? gp[1] : gn[1]);
    const float e2 = role == 0 ? ga[2] : (role == 1 ? gp[2] : gn[2]);
    const float dot = me.y[0] * e0 + me.y[1] * e1 + me.y[2] * e2;
    r.q0 = (e0 - me.y[0] * dot) / me.den;
    r.q1 = (e1 - me.y[1] * dot) / me.den;
    r.q2 = (e2 - me.y[2] * dot) / me.den;
    return r;
}

// tile_gemm with the A operand made by the caller: araw(q) is fetched a block of steps ahead (a global or LDS read), amake(raw, q)
// turns it into the operand when it is used
template <class AR, class AM>
__device__ __forceinline__ void tile_gemm_fn(int Q, const float* __restrict__ Bm, int ldb, bool aok, int lane, int wave, f32x16& acc,
                                             float& asum, AR&& araw, AM&& amake) {
    constexpr int UB = 8;
    const int half = lane >> 5, l31 = lane & 31;
    const float* bp = Bm + l31;
    const int steps = (Q + 1) / 2;
    const int mine = (steps - wave + 3) / 4;
    float ab[2][UB], bb[2][UB];
    auto fetch = [&](int buf, int i0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int q = 2 * (wave + 4 * (i0 + u)) + half;
            const int qc = q < Q ? q : Q - 1;
            ab[buf][u] = araw(qc);
            bb[buf][u] = bp[(size_t)qc * ldb];
        }
    };
    auto mul = [&](int buf, int i0) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int q = 2 * (wave + 4 * (i0 + u)) + half;
            const int qc = q < Q ? q : Q - 1;
            const float a = (aok && q < Q && i0 + u < mine) ? amake(ab[buf][u], qc) : 0.f;
            asum += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[buf][u], acc, 0, 0, 0);
        }
    };
    fetch(0, 0);
#pragma unroll 1
    for (int i0 = 0; i0 < mine; i0 += 2 * UB) {
        fetch(1, i0 + UB);
        __builtin_amdgcn_sched_barrier(0);
        mul(0, i0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, i0 + 2 * UB);
        __builtin_amdgcn_sched_barrier(0);
        mul(1, i0 + UB);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// rows [0, R) of this call are rows [g0, g0 + R) of the (3B) stack [image | positive | negative]; hid, x, gx belong to this call's rows
template <int K>
__global__ void __launch_bounds__(256)
heads_bwd_fused_kernel(int R, int B, int g0, const float* part, const float* __restrict__ b1_img, const float* __restrict__ b1_snd,
                       float margin, float inv_count, const float* __restrict__ x, const float* __restrict__ w0,
                       const float* __restrict__ w1, const float* __restrict__ hid, float* __restrict__ dw0, float* __restrict__ db0,
                       float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ gx, float* __restrict__ emb_raw,
                       float* __restrict__ emb, float* __restrict__ loss_out) {
    constexpr int KB = K / 32;
    __shared__ float red[4][1024];
    __shared__ float reda[4][32];
    __shared__ float4 qs[kFusedMaxRows];                       // (q0, q1, q2, -) of the rows this tile reduces over
    __shared__ float ht[32][kHid + 1];                         // a data-gradient tile's 32 rows of hid
    __shared__ float w1s[3][kHid];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x;
    const int n_dw0 = 4 * KB, n_dw1 = 4;
    const bool loss_blk = loss_out && tile == (int)gridDim.x - 1;
    const int kind = loss_blk ? 3 : tile < n_dw0 ? 0 : tile < n_dw0 + n_dw1 ? 1 : 2;
    int i0 = 0, i1 = 0;
    if (kind == 0) { i0 = tile / KB; i1 = tile - i0 * KB; }                      // nb, kb
    else if (kind == 1) i0 = tile - n_dw0;                                     // nb
    else if (kind == 2) { const int g = tile - n_dw0 - n_dw1; i0 = g / KB; i1 = g - i0 * KB; }      // rb, kb
    // ---- the rows' scalars ----
    const int rfirst = kind == 2 ? i0 * 32 : 0, rlast = kind == 2 ? (i0 * 32 + 32 < R ? i0 * 32 + 32 : R) : R;
    float local = 0.f;
    for (int r = rfirst + tid; r < rlast; r += 256) {
        Emb3 me;
        const RowQ q = row_scalars(g0 + r, B, part, b1_img, b1_snd, margin, inv_count, me);
        qs[r - rfirst] = make_float4(q.q0, q.q1, q.q2, 0.f);
        local += q.lterm;                                      // (i = tid, tid + 256, ...: triplet_loss_kernel's order)
        if (kind == 1 && i0 == 0) {
            const int g = g0 + r;
#pragma unroll
            for (int d = 0; d < 3; ++d) { emb_raw[g * 3 + d] = me.raw[d]; emb[g * 3 + d] = me.y[d]; }
        }
    }
    if (kind == 3) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
        if (lane == 0) reda[0][wave] = local;
        __syncthreads();
        if (tid == 0) loss_out[0] = ((reda[0][0] + reda[0][1]) + (reda[0][2] + reda[0][3])) * inv_count;
        return;
    }
    if (kind == 2) {
        for (int e = tid; e < 32 * kHid / 4; e += 256) {
            const int r = e / (kHid / 4), c4 = e - r * (kHid / 4);
            const int row = i0 * 32 + r;
            const float4 v = row < R ? ((const float4*)(hid + (size_t)row * kHid))[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            ht[r][4 * c4] = v.x; ht[r][4 * c4 + 1] = v.y; ht[r][4 * c4 + 2] = v.z; ht[r][4 * c4 + 3] = v.w;
        }
        for (int e = tid; e < 3 * kHid; e += 256) w1s[e / kHid][e % kHid] = w1[e];
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    if (kind == 0) {
        const int n = i0 * 32 + l31;
        const float wa = w1[n], wb = w1[kHid + n], wc = w1[2 * kHid + n];
        const float* hp = hid + n;
        tile_gemm_fn(R, x + i1 * 32, K, true, lane, wave, acc, asum,
                     [&](int q) { return hp[(size_t)q * kHid]; },
                     [&](float h, int q) { const float4 s = qs[q]; const float v = s.x * wa + s.y * wb + s.z * wc; return h > 0.f ? v : 0.f; });
    } else if (kind == 1) {
        tile_gemm_fn(R, hid + i0 * 32, kHid, l31 < 3, lane, wave, acc, asum,
                     [&](int q) { return ((const float*)&qs[q])[l31 < 3 ? l31 : 0]; },
                     [&](float v, int) { return v; });
    } else {
        const bool aok = l31 < R - i0 * 32;
        const float4 s = qs[aok ? l31 : 0];
        tile_gemm_fn(kHid, w0 + i1 * 32, K, aok, lane, wave, acc, asum,
                     [&](int q) { return ht[l31][q]; },
                     [&](float h, int q) { const float v = s.x * w1s[0][q] + s.y * w1s[1][q] + s.z * w1s[2][q]; return h > 0.f ? v : 0.f; });
    }
    // fold the four K slices (fixed order)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
    asum += __shfl_down(asum, 32, 64);
    if (half == 0) reda[wave][l31] = asum;
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        const int i = e >> 5, j = e & 31;                // D[i][j]
        if (kind == 0) {
            dw0[(size_t)(i0 * 32 + i) * K + i1 * 32 + j] = v;
        } else if (kind == 1) {
            if (i < 3) dw1[i * kHid + i0 * 32 + j] = v;
        } else {
            const int row = i0 * 32 + i;
            if (row < R) {
                const size_t o = (size_t)row * K + i1 * 32 + j;
                gx[o] = x[o] > 0.f ? v : 0.f;
            }
        }
    }
    if (tid < 32) {
        const float v = (reda[0][tid] + reda[1][tid]) + (reda[2][tid] + reda[3][tid]);
        if (kind == 0 && i1 == 0) db0[i0 * 32 + tid] = v;
        if (kind == 1 && i0 == 0 && tid < 3) db1[tid] = v;
    }
}
}  // namespace

// ------------------------------------------------------------------------------------------
// forward, spread over 4x the workgroups: grid = (row blocks of 32, 4 blocks of 32 hidden units); 4 waves split K.
// Each workgroup leaves its 32 hidden columns (post-ReLU, backward needs them) and its PARTIAL of the 128 -> 3
// output layer; heads_finish_kernel adds the four partials in a fixed order, the bias, and normalises.
// (One 16-wave workgroup per 32 rows would keep only 8 / 16 CUs busy for 8 us of matrix work each.)
// ------------------------------------------------------------------------------------------
// Small batches (one row block: the RL stage's 8 envs, where every launch is at the floor of what a launch costs): the LAST of the
// four workgroups to arrive also runs heads_finish_kernel's arithmetic on the block's rows and, when asked, the reward's row dot
// (var_set_reward_dot) -- two launches fewer on the frozen encoder's latency path.  The partials travel as agent-scope stores /
// loads then (var_common.h: join_store / join_load), the arrival count rewinds itself.
struct FinishArgs {
    const float* b1; float* emb_raw; float* emb; float* out0; float* out1; int split;
    unsigned* ctr;                 // nullptr: no finish in this launch
    const float* dot_with; float* dot_out;
};

template <int K>
__global__ void __launch_bounds__(256)
heads_fwd_split_kernel(const float* __restrict__ x, int R, const float* __restrict__ w0t, const float* __restrict__ b0,
                       const float* __restrict__ w1, float* __restrict__ hid, float* __restrict__ part, unsigned* sig, unsigned* sig_other,
                       FinishArgs fin) {
    constexpr int LDX = K + 1, NT = 256, KSL = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                    // [32][K+1]
    float* ps = lds;                    // reused after the GEMM: partial tiles [4][32][33]
    float* hs = lds + 4 * 32 * 33;      // [32][33]
    const int tid = threadIdx.x, lane = tid & 63, ksl = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = blockIdx.y;
    const int r0 = blockIdx.x * 32;
    {
        constexpr int TOT = 32 * K / 4;
        const float4* src = (const float4*)(x + (size_t)r0 * K);
        const int lim = (R - r0) * (K / 4);
        constexpr int IT = (TOT + NT - 1) / NT;
        float4 v[IT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int e = tid + NT * u;
            v[u] = (e < TOT && e < lim) ? src[e] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int e = tid + NT * u;
            if (e < TOT) {
                const int r = (e * 4) / K, k = (e * 4) - r * K;
                float* d = xs + r * LDX + k;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int STEPS = K / 2 / KSL;
    constexpr int UF = (STEPS % 8 == 0) ? 8 : 10;
    static_assert((K / 2) % KSL == 0 && STEPS % UF == 0, "K must split into 4 slices of whole prefetch blocks");
    constexpr int NBK = STEPS / UF;
    const int kbase = ksl * STEPS;
    const float* wl = w0t + (size_t)(2 * kbase + half) * kHid + nblk * 32 + l31;
    const float* xl = xs + l31 * LDX + 2 * kbase + half;
    float wb[2][UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) wb[0][u] = wl[(2 * u) * kHid];
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk) {
        if (blk + 1 < NBK) {
#pragma unroll
            for (int u = 0; u < UF; ++u) wb[(blk + 1) & 1][u] = wl[(2 * ((blk + 1) * UF + u)) * kHid];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UF; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xl[2 * (blk * UF + u)], wb[blk & 1][u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                    // everyone is done reading xs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        ps[(ksl * 32 + row) * 33 + l31] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += NT) {
        const int row = e >> 5, n = e & 31;
        float v = ((ps[row * 33 + n] + ps[(32 + row) * 33 + n]) + (ps[(64 + row) * 33 + n] + ps[(96 + row) * 33 + n])) + b0[nblk * 32 + n];
        v = v > 0.f ? v : 0.f;
        hs[row * 33 + n] = v;
        if (r0 + row < R) hid[(size_t)(r0 + row) * kHid + nblk * 32 + n] = v;
    }
    __syncthreads();
    if (tid < 96) {
        const int r = tid / 3, d = tid - r * 3;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) s += hs[r * 33 + j] * w1[d * kHid + nblk * 32 + j];
        if (r0 + r < R) {
            if (sig || fin.ctr) join_store(part + ((size_t)(r0 + r) * 4 + nblk) * 4 + d, s);
            else part[((size_t)(r0 + r) * 4 + nblk) * 4 + d] = s;
        }
    }
    if (sig) join_signal(sig, gridDim.x * gridDim.y, sig_other);   // (training step: hand-over between the streams on the device)
    if (fin.ctr) {
        __shared__ int last_s;
        __syncthreads();                                       // (every join_store of this workgroup is acknowledged)
        if (tid == 0) {
            last_s = atomicAdd(fin.ctr, 1u) == gridDim.y - 1;
            if (last_s) atomicExch(fin.ctr, 0u);
        }
        __syncthreads();
        const int row = r0 + tid;
        if (last_s && tid < 32 && row < R) {
            float p[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) p[k] = join_load(part + (size_t)row * 16 + k);
            // (heads_finish_kernel's arithmetic)
            const float a = ((p[0] + p[4]) + (p[8] + p[12])) + fin.b1[0];
            const float b = ((p[1] + p[5]) + (p[9] + p[13])) + fin.b1[1];
            const float c = ((p[2] + p[6]) + (p[10] + p[14])) + fin.b1[2];
            const float nrm = sqrtf(a * a + b * b + c * c);
            const float den = nrm > 1e-12f ? nrm : 1e-12f;
            const float y0 = a / den, y1 = b / den, y2 = c / den;
            fin.emb_raw[row * 3 + 0] = a; fin.emb_raw[row * 3 + 1] = b; fin.emb_raw[row * 3 + 2] = c;
            fin.emb[row * 3 + 0] = y0; fin.emb[row * 3 + 1] = y1; fin.emb[row * 3 + 2] = y2;
            float* o = row < fin.split ? (fin.out0 ? fin.out0 + 3 * row : nullptr) : (fin.out1 ? fin.out1 + 3 * (row - fin.split) : nullptr);
            if (o) { o[0] = y0; o[1] = y1; o[2] = y2; }
            if (fin.dot_out) {                                   // row_dot_kernel's sum, k = 0, 1, 2
                float sdot = 0.f;
                sdot += y0 * fin.dot_with[row * 3 + 0];
                sdot += y1 * fin.dot_with[row * 3 + 1];
                sdot += y2 * fin.dot_with[row * 3 + 2];
                fin.dot_out[row] = sdot;
            }
        }
    }
}

// emb_raw = b1 + ((part0 + part1) + (part2 + part3)); emb = emb_raw / max(||emb_raw||, 1e-12)
__global__ void __launch_bounds__(256)
heads_finish_kernel(const float* __restrict__ part, const float* __restrict__ b1, int R, float* __restrict__ emb_raw,
                    float* __restrict__ emb, float* __restrict__ out0, float* __restrict__ out1, int split) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= R) return;
    const float4* p = (const float4*)(part + (size_t)row * 16);
    const float4 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
    const float a = ((p0.x + p1.x) + (p2.x + p3.x)) + b1[0];
    const float b = ((p0.y + p1.y) + (p2.y + p3.y)) + b1[1];
    const float c = ((p0.z + p1.z) + (p2.z + p3.z)) + b1[2];
    const float nrm = sqrtf(a * a + b * b + c * c);
    const float den = nrm > 1e-12f ? nrm : 1e-12f;
    emb_raw[row * 3 + 0] = a; emb_raw[row * 3 + 1] = b; emb_raw[row * 3 + 2] = c;
    emb[row * 3 + 0] = a / den; emb[row * 3 + 1] = b / den; emb[row * 3 + 2] = c / den;
    // the caller's copy of the normalised embedding (var_arm_encoder_fwd): rows [0, split) -> out0, the rest -> out1
    float* o = row < split ? (out0 ? out0 + 3 * row : nullptr) : (out1 ? out1 + 3 * (row - split) : nullptr);
    if (o) { o[0] = a / den; o[1] = b / den; o[2] = c / den; }
}

template <int K>
static int run_heads_fwd(var_ctx* c, hipStream_t s, const float* x, int R, const float* w0t, const float* b0,
                         const float* w1, const float* b1, float* hid, float* emb_raw, float* emb, float* part,
                         bool finish, float* out0 = nullptr, float* out1 = nullptr, int split = 0, unsigned* sig = nullptr,
                         unsigned* sig_other = nullptr) {
    constexpr int A1 = 32 * (K + 1) * 4, A2 = 5 * 32 * 33 * 4;
    constexpr int LDS_BYTES = A1 > A2 ? A1 : A2;
    static unsigned attr_set = 0;      // bit d: set on device d (function attributes are per device)
    if (!(attr_set & var_dev_bit(c))) {
        VAR_HIP_CHECK(c, hipFuncSetAttribute((const void*)heads_fwd_split_kernel<K>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set |= var_dev_bit(c);
    }
    // one row block and a finish wanted: the last workgroup finishes (FinishArgs above); the reward's row dot rides along when
    // var_set_reward_dot armed it for this (image) head
    const bool fin_in = finish && R <= 32 && !sig;
    FinishArgs fin{b1, emb_raw, emb, out0, out1, split, fin_in ? c->jsig + 12 : nullptr, nullptr, nullptr};
    if (fin_in && K == kImgFeat && c->dot_out) { fin.dot_with = c->dot_with; fin.dot_out = c->dot_out; }
    if (K == kImgFeat) { c->dot_with = nullptr; c->dot_out = nullptr; }          // (armed for one forward)
    hipLaunchKernelGGL(heads_fwd_split_kernel<K>, dim3((R + 31) / 32, 4), dim3(256), LDS_BYTES, s, x, R, w0t, b0, w1, hid, part, sig, sig_other, fin);
    if (finish && !fin_in) hipLaunchKernelGGL(heads_finish_kernel, dim3((R + 255) / 256), dim3(256), 0, s, part, b1, R, emb_raw, emb, out0, out1, split);
    return VAR_OK;
}

int launch_heads_fwd(var_ctx* c, hipStream_t s, hipStream_t ss, const float* params, int B, bool has_img,
                     bool has_pos, bool has_neg, bool finish) {
    const ParamLayout& L = c->pl;
    const PackLayout& K = c->kl;
    int rc;
    if (has_img && c->head_in_mid) {
        // the fused conv 3-5 kernel already left hid_i and the 128 -> 3 partials of every image -- and, when a finish was wanted
        // (var_ctx::mid_finish), the embeddings themselves
        if (finish && !c->mid_finish) hipLaunchKernelGGL(heads_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, s, c->head_part,
                                                         params + L.ih_b1, B, c->emb_raw, c->emb, c->out_img, (float*)nullptr, B);
    } else if (has_img) {
        ProfScope prof(c, s, TAG_HEADS_FWD);
        if ((rc = run_heads_fwd<kImgFeat>(c, s, c->act[5], B, c->wpack + K.ih_w0t, params + L.ih_b0, params + L.ih_w1,
                                          params + L.ih_b1, c->hid_i, c->emb_raw, c->emb, c->head_part, finish, c->out_img, nullptr, B,
                                          c->dev_join ? c->jsig + 4 : nullptr, c->jsig)) != VAR_OK) return rc;
    }
    if (has_pos || has_neg) {
        const int lo = has_pos ? 0 : B, hi = has_neg ? 2 * B : B;
        if ((rc = run_heads_fwd<kSndFeat>(c, ss, c->sact[4] + (size_t)lo * kSndFeat, hi - lo, c->wpack + K.sh_w0t,
                                          params + L.sh_b0, params + L.sh_w1, params + L.sh_b1,
                                          c->hid_s + (size_t)lo * kHid, c->emb_raw + 3 * (B + lo),
                                          c->emb + 3 * (B + lo), c->head_part + 16 * (size_t)(B + lo), finish,
                                          lo == 0 ? c->out_pos : nullptr, c->out_neg, lo == 0 ? B : 0,
                                          c->dev_join ? c->jsig : nullptr, c->jsig + 4)) != VAR_OK) return rc;
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

int launch_triplet_loss(var_ctx* c, hipStream_t s, const float* params, int B, float margin, float inv_count,
                        float* loss_out) {
    const ParamLayout& L = c->pl;
    ProfScope prof(c, s, TAG_TRIPLET);
    hipLaunchKernelGGL(triplet_loss_kernel, dim3(1), dim3(256), 0, s, c->head_part, params + L.ih_b1, params + L.sh_b1, B,
                       margin, inv_count, loss_out);
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}

// gemb (3B,3) must hold the gradients wrt the normalised embeddings [img | pos | neg].
// Produces gact[5] (B,576), gsact[4] (2B,160) and the 8 head gradient tensors.
int launch_heads_bwd(var_ctx* c, hipStream_t s, hipStream_t ss, const float* params, float* grads, int B, bool has_img,
                     int snd_lo, int snd_hi, bool fused, float margin, float inv_count, float* loss_out) {
    const ParamLayout& L = c->pl;
    const size_t mB = (size_t)c->maxB;
    float* graw = c->gemb + 9 * mB;                       // (3B,4)
    float* lterms = (fused && has_img && loss_out) ? c->gemb + 21 * mB : nullptr;     // (B): per-sample loss terms (fused step)
    float* ghidT = c->ghid + 3 * mB * kHid;               // second half of the ghid buffer: [128][rows]
    // The training step (fused): rows and products of a head in ONE launch each (heads_bwd_fused_kernel)
    if (fused && B <= kFusedMaxRows && snd_hi - snd_lo <= kFusedMaxRows) {
        ProfScope prof(c, s, TAG_HEADS_BWD_W);
        if (has_img) {
            const int tiles = 4 * (kImgFeat / 32) + 4 + ((B + 31) / 32) * (kImgFeat / 32);
            hipLaunchKernelGGL(heads_bwd_fused_kernel<kImgFeat>, dim3(tiles + (loss_out ? 1 : 0)), dim3(256), 0, s, B, B, 0,
                               (const float*)c->head_part, params + L.ih_b1, params + L.sh_b1, margin, inv_count,
                               (const float*)c->act[5], params + L.ih_w0, params + L.ih_w1, (const float*)c->hid_i,
                               grads + L.ih_w0, grads + L.ih_b0, grads + L.ih_w1, grads + L.ih_b1, c->gact[5], c->emb_raw, c->emb,
                               loss_out);
        }
        if (snd_hi > snd_lo) {
            const int R = snd_hi - snd_lo;
            const int tiles = 4 * (kSndFeat / 32) + 4 + ((R + 31) / 32) * (kSndFeat / 32);
            hipLaunchKernelGGL(heads_bwd_fused_kernel<kSndFeat>, dim3(tiles), dim3(256), 0, ss, R, B, B + snd_lo,
                               (const float*)c->head_part, params + L.ih_b1, params + L.sh_b1, margin, inv_count,
                               (const float*)(c->sact[4] + (size_t)snd_lo * kSndFeat), params + L.sh_w0, params + L.sh_w1,
                               (const float*)(c->hid_s + (size_t)snd_lo * kHid), grads + L.sh_w0, grads + L.sh_b0, grads + L.sh_w1,
                               grads + L.sh_b1, c->gsact[4] + (size_t)snd_lo * kSndFeat, c->emb_raw, c->emb, (float*)nullptr);
        }
        VAR_HIP_CHECK(c, hipGetLastError());
        return VAR_OK;
    }
    {
        // fused: the rows finish their sample's embeddings and form the triplet gradient themselves (forward
        // partials in c->head_part); otherwise gemb holds the gradients wrt the normalised embeddings
        ProfScope prof(c, s, TAG_HEADS_BWD_ROWS);
        if (has_img) {
            if (fused)
                hipLaunchKernelGGL(heads_bwd_rows_fused_kernel, dim3(B), dim3(128), 0, s, B, B, 0, params + L.ih_w1,
                                   c->hid_i, c->head_part, params + L.ih_b1, params + L.sh_b1, margin, inv_count,
                                   c->emb_raw, c->emb, graw, c->ghid, ghidT, c->dev_join ? c->jsig : nullptr, 6, lterms);
            else
                hipLaunchKernelGGL(heads_bwd_rows_kernel, dim3(B), dim3(128), 0, s, B, params + L.ih_w1, c->hid_i,
                                   c->emb_raw, c->emb, c->gemb, graw, c->ghid, ghidT);
        }
        if (snd_hi > snd_lo) {
            const int R = snd_hi - snd_lo;
            if (fused)
                hipLaunchKernelGGL(heads_bwd_rows_fused_kernel, dim3(R), dim3(128), 0, ss, R, B, B + snd_lo,
                                   params + L.sh_w1, c->hid_s + (size_t)snd_lo * kHid, c->head_part, params + L.ih_b1,
                                   params + L.sh_b1, margin, inv_count, c->emb_raw, c->emb, graw + 4 * (B + snd_lo),
                                   c->ghid + (size_t)(B + snd_lo) * kHid, ghidT + (size_t)B * kHid, c->dev_join ? c->jsig + 4 : nullptr, 1,
                                   (float*)nullptr);
            else
                hipLaunchKernelGGL(heads_bwd_rows_kernel, dim3(R), dim3(128), 0, ss, R, params + L.sh_w1,
                                   c->hid_s + (size_t)snd_lo * kHid, c->emb_raw + 3 * (B + snd_lo),
                                   c->emb + 3 * (B + snd_lo), c->gemb + 3 * (B + snd_lo), graw + 4 * (B + snd_lo),
                                   c->ghid + (size_t)(B + snd_lo) * kHid, ghidT + (size_t)B * kHid);
        }
    }
    {
        ProfScope prof(c, s, TAG_HEADS_BWD_W);
        if (has_img) {
            const int tiles = 4 * (kImgFeat / 32) + 4 + ((B + 31) / 32) * (kImgFeat / 32);
            hipLaunchKernelGGL(heads_bwd_gemm_kernel<kImgFeat>, dim3(tiles + (lterms ? 1 : 0)), dim3(256), 0, s, B, B,
                               c->act[5], params + L.ih_w0, c->hid_i, graw, c->ghid, ghidT,
                               grads + L.ih_w0, grads + L.ih_b0, grads + L.ih_w1, grads + L.ih_b1, c->gact[5],
                               (const float*)lterms, inv_count, loss_out);
        }
        if (snd_hi > snd_lo) {
            const int R = snd_hi - snd_lo;
            const int tiles = 4 * (kSndFeat / 32) + 4 + ((R + 31) / 32) * (kSndFeat / 32);
            hipLaunchKernelGGL(heads_bwd_gemm_kernel<kSndFeat>, dim3(tiles), dim3(256), 0, ss, R, R,
                               c->sact[4] + (size_t)snd_lo * kSndFeat, params + L.sh_w0,
                               c->hid_s + (size_t)snd_lo * kHid, graw + 4 * (B + snd_lo),
                               c->ghid + (size_t)(B + snd_lo) * kHid, ghidT + (size_t)B * kHid,
                               grads + L.sh_w0, grads + L.sh_b0, grads + L.sh_w1, grads + L.sh_b1,
                               c->gsact[4] + (size_t)snd_lo * kSndFeat, (const float*)nullptr, 0.f, (float*)nullptr);
        }
    }
    VAR_HIP_CHECK(c, hipGetLastError());
    return VAR_OK;
}
