/*
 * ORACLE -- test infrastructure only.  Nothing in the product path may call,
 * link or load this file; it is the CPU checker that tests/, smoke() and
 * bench.py's cpu_baseline leg compare the HIP path against.
 *
 * Plain-C restatement (float storage, double accumulation) of the VAR
 * contrastive-pretext step of PeixinC/VoiceControlledRobot-VAR, Kuka model:
 *
 *   image CNN    models/pretext/arm_pretext_model.py:9-18   5x[Conv3x3 s2 p1 + ReLU] + Flatten
 *   sound CNN    models/pretext/arm_pretext_model.py:21-34  Conv(1,32,(5,40),s(2,1)) + 3x Conv(32,32,(3,1),s(2,1)), ReLUs
 *   heads        models/pretext/arm_pretext_model.py:46-56  Linear(576|160,128)+ReLU+Linear(128,3)
 *   routing      models/pretext/pretext_base.py:10-41       image[:, :3]; F.normalize(p=2,dim=1,eps=1e-12)
 *   loss         VAR/pretext_VAR.py:38,64                   TripletMarginLoss(margin=1, p=2, eps=1e-6), mean
 *   backward     VAR/pretext_VAR.py:68                      autograd of the above
 *   optimiser    VAR/pretext_VAR.py:33-35,69                Adam(lr, betas=(.9,.999), eps=1e-8, weight_decay) (coupled L2)
 *   image scale  dataset.py:67-68                           u8 -> f32 division by 255.
 *
 * Pinned by tests/test_oracle_c.py against the .npz files in tests/golden, which were
 * produced by importing the reference classes (tests/golden/make_golden.py).
 *
 * Parameter arena = the 26 tensors of state_dict() in registration order,
 * each in its PyTorch layout (OIHW / (out,in)), 213478 floats.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IMG_C0 3
#define N_IMG_CONV 5
#define N_SND_CONV 4
#define EMB 3
#define HID 128
#define SND_T 100
#define SND_F 40

static const int IMG_CH[6] = {3, 32, 32, 64, 64, 64};
static const int SND_CH[5] = {1, 32, 32, 32, 32};
static const int SND_KH[4] = {5, 3, 3, 3};
static const int SND_KW[4] = {40, 1, 1, 1};

typedef struct {
    /* offsets (in floats) into the flat parameter arena */
    int img_w[5], img_b[5];
    int snd_w[4], snd_b[4];
    int ih_w0, ih_b0, ih_w1, ih_b1;
    int sh_w0, sh_b0, sh_w1, sh_b1;
    int total;
    int img_feat;   /* 64*3*3 */
    int snd_feat;   /* 32*5   */
} layout_t;

static int out_sz(int h) { return (h + 2 - 3) / 2 + 1; }

static void make_layout(layout_t *L) {
    int o = 0;
    for (int i = 0; i < 5; i++) {
        L->img_w[i] = o; o += IMG_CH[i + 1] * IMG_CH[i] * 9;
        L->img_b[i] = o; o += IMG_CH[i + 1];
    }
    for (int i = 0; i < 4; i++) {
        L->snd_w[i] = o; o += SND_CH[i + 1] * SND_CH[i] * SND_KH[i] * SND_KW[i];
        L->snd_b[i] = o; o += SND_CH[i + 1];
    }
    L->img_feat = 64 * 9;
    L->snd_feat = 32 * 5;
    L->ih_w0 = o; o += HID * L->img_feat; L->ih_b0 = o; o += HID;
    L->ih_w1 = o; o += EMB * HID;         L->ih_b1 = o; o += EMB;
    L->sh_w0 = o; o += HID * L->snd_feat; L->sh_b0 = o; o += HID;
    L->sh_w1 = o; o += EMB * HID;         L->sh_b1 = o; o += EMB;
    L->total = o;
}

int orc_param_count(void) { layout_t L; make_layout(&L); return L.total; }

/* ---------------- generic conv (NCHW, OIHW), fused bias + ReLU ---------------- */
static void conv_fwd(const float *x, const float *w, const float *b, float *y,
                     int B, int Ci, int H, int W, int Co, int KH, int KW, int SH, int SW, int PH, int PW_) {
    int Ho = (H + 2 * PH - KH) / SH + 1, Wo = (W + 2 * PW_ - KW) / SW + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; n++)
        for (int co = 0; co < Co; co++) {
            for (int oy = 0; oy < Ho; oy++)
                for (int ox = 0; ox < Wo; ox++) {
                    double acc = b[co];
                    for (int ci = 0; ci < Ci; ci++)
                        for (int ky = 0; ky < KH; ky++) {
                            int iy = oy * SH + ky - PH;
                            if (iy < 0 || iy >= H) continue;
                            const float *xr = x + (((size_t)n * Ci + ci) * H + iy) * W;
                            const float *wr = w + (((size_t)co * Ci + ci) * KH + ky) * KW;
                            for (int kx = 0; kx < KW; kx++) {
                                int ix = ox * SW + kx - PW_;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)xr[ix] * (double)wr[kx];
                            }
                        }
                    float v = (float)acc;
                    y[(((size_t)n * Co + co) * Ho + oy) * Wo + ox] = v > 0.f ? v : 0.f;
                }
        }
}

/* dy is the gradient wrt the post-ReLU output y; masks by y>0 in place, then
 * dw += , db +=, dx = (dx may be NULL). */
static void conv_bwd(const float *x, const float *w, const float *y, float *dy,
                     float *dx, float *dw, float *db,
                     int B, int Ci, int H, int W, int Co, int KH, int KW, int SH, int SW, int PH, int PW_) {
    int Ho = (H + 2 * PH - KH) / SH + 1, Wo = (W + 2 * PW_ - KW) / SW + 1;
    size_t ny = (size_t)B * Co * Ho * Wo;
    for (size_t i = 0; i < ny; i++) if (!(y[i] > 0.f)) dy[i] = 0.f;
    /* weight + bias grads: parallel over output channel (no races) */
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Co; co++) {
        double sb = 0.0;
        for (int n = 0; n < B; n++) {
            const float *g = dy + ((size_t)n * Co + co) * Ho * Wo;
            for (int i = 0; i < Ho * Wo; i++) sb += g[i];
        }
        db[co] = (float)sb;
        for (int ci = 0; ci < Ci; ci++)
            for (int ky = 0; ky < KH; ky++)
                for (int kx = 0; kx < KW; kx++) {
                    double s = 0.0;
                    for (int n = 0; n < B; n++) {
                        const float *g = dy + ((size_t)n * Co + co) * Ho * Wo;
                        const float *xi = x + ((size_t)n * Ci + ci) * H * W;
                        for (int oy = 0; oy < Ho; oy++) {
                            int iy = oy * SH + ky - PH;
                            if (iy < 0 || iy >= H) continue;
                            for (int ox = 0; ox < Wo; ox++) {
                                int ix = ox * SW + kx - PW_;
                                if (ix < 0 || ix >= W) continue;
                                s += (double)g[oy * Wo + ox] * (double)xi[iy * W + ix];
                            }
                        }
                    }
                    dw[(((size_t)co * Ci + ci) * KH + ky) * KW + kx] = (float)s;
                }
    }
    if (!dx) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; n++)
        for (int ci = 0; ci < Ci; ci++) {
            double *acc = (double *)calloc((size_t)H * W, sizeof(double));
            for (int co = 0; co < Co; co++) {
                const float *g = dy + ((size_t)n * Co + co) * Ho * Wo;
                const float *wr = w + ((size_t)co * Ci + ci) * KH * KW;
                for (int oy = 0; oy < Ho; oy++)
                    for (int ky = 0; ky < KH; ky++) {
                        int iy = oy * SH + ky - PH;
                        if (iy < 0 || iy >= H) continue;
                        for (int ox = 0; ox < Wo; ox++) {
                            double gv = g[oy * Wo + ox];
                            if (gv == 0.0) continue;
                            for (int kx = 0; kx < KW; kx++) {
                                int ix = ox * SW + kx - PW_;
                                if (ix < 0 || ix >= W) continue;
                                acc[iy * W + ix] += gv * (double)wr[ky * KW + kx];
                            }
                        }
                    }
            }
            float *d = dx + ((size_t)n * Ci + ci) * H * W;
            for (int i = 0; i < H * W; i++) d[i] = (float)acc[i];
            free(acc);
        }
}

/* y[B,N] = x[B,K] @ w[N,K]^T + b, optional ReLU */
static void linear_fwd(const float *x, const float *w, const float *b, float *y, int B, int K, int N, int relu) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; i++)
        for (int n = 0; n < N; n++) {
            double acc = b[n];
            for (int k = 0; k < K; k++) acc += (double)x[(size_t)i * K + k] * (double)w[(size_t)n * K + k];
            float v = (float)acc;
            y[(size_t)i * N + n] = (relu && !(v > 0.f)) ? 0.f : v;
        }
}

/* dy wrt post-activation output; masked in place when relu. dw/db are ACCUMULATED (+=) */
static void linear_bwd(const float *x, const float *w, const float *y, float *dy, float *dx, float *dw, float *db,
                       int B, int K, int N, int relu) {
    if (relu) for (size_t i = 0; i < (size_t)B * N; i++) if (!(y[i] > 0.f)) dy[i] = 0.f;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        double sb = 0.0;
        for (int i = 0; i < B; i++) sb += dy[(size_t)i * N + n];
        db[n] += (float)sb;
        for (int k = 0; k < K; k++) {
            double s = 0.0;
            for (int i = 0; i < B; i++) s += (double)dy[(size_t)i * N + n] * (double)x[(size_t)i * K + k];
            dw[(size_t)n * K + k] += (float)s;
        }
    }
    if (!dx) return;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; i++)
        for (int k = 0; k < K; k++) {
            double s = 0.0;
            for (int n = 0; n < N; n++) s += (double)dy[(size_t)i * N + n] * (double)w[(size_t)n * K + k];
            dx[(size_t)i * K + k] = (float)s;
        }
}

/* F.normalize(x, p=2, dim=1, eps=1e-12) on (B,3) */
void orc_normalize_fwd(const float *x, float *y, float *nrm, int B) {
    for (int i = 0; i < B; i++) {
        double s = 0.0;
        for (int d = 0; d < EMB; d++) s += (double)x[i * EMB + d] * (double)x[i * EMB + d];
        float n = (float)sqrt(s);
        float den = n > 1e-12f ? n : 1e-12f;
        for (int d = 0; d < EMB; d++) y[i * EMB + d] = x[i * EMB + d] / den;
        if (nrm) nrm[i] = den;
    }
}

void orc_normalize_bwd(const float *y, const float *nrm, const float *gy, float *gx, int B) {
    for (int i = 0; i < B; i++) {
        double dot = 0.0;
        for (int d = 0; d < EMB; d++) dot += (double)y[i * EMB + d] * (double)gy[i * EMB + d];
        for (int d = 0; d < EMB; d++)
            gx[i * EMB + d] = (float)(((double)gy[i * EMB + d] - (double)y[i * EMB + d] * dot) / (double)nrm[i]);
    }
}

/* TripletMarginLoss(margin, p=2, eps=1e-6, swap=False, reduction='mean');
 * pairwise_distance adds eps to the DIFFERENCE.  Writes grads when ga != NULL. */
float orc_triplet(const float *a, const float *p, const float *n, int B, float margin,
                  float *ga, float *gp, float *gn) {
    const double eps = 1e-6;
    double total = 0.0;
    for (int i = 0; i < B; i++) {
        double dp[EMB], dn[EMB], sp = 0.0, sn = 0.0;
        for (int d = 0; d < EMB; d++) {
            dp[d] = (double)(float)((float)(a[i * EMB + d] - p[i * EMB + d]) + (float)eps);
            dn[d] = (double)(float)((float)(a[i * EMB + d] - n[i * EMB + d]) + (float)eps);
            sp += dp[d] * dp[d]; sn += dn[d] * dn[d];
        }
        double dap = sqrt(sp), dan = sqrt(sn);
        double l = dap - dan + (double)margin;
        int active = l > 0.0;
        if (active) total += l;
        if (ga) {
            double s = active ? 1.0 / B : 0.0;
            for (int d = 0; d < EMB; d++) {
                double up = dap > 0.0 ? dp[d] / dap : 0.0;
                double un = dan > 0.0 ? dn[d] / dan : 0.0;
                ga[i * EMB + d] = (float)(s * (up - un));
                gp[i * EMB + d] = (float)(-s * up);
                gn[i * EMB + d] = (float)(s * un);
            }
        }
    }
    return (float)(total / B);
}

/* torch.optim.Adam single-tensor step (amsgrad=False, maximize=False), step t >= 1 */
void orc_adam(float *p, const float *g, float *m, float *v, long n, float lr, float b1, float b2,
              float eps, float wd, int t) {
    double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
    double step_size = (double)lr / bc1;
    double bc2s = sqrt(bc2);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        float gi = g[i] + wd * p[i];
        m[i] = b1 * m[i] + (1.f - b1) * gi;
        v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
        double denom = sqrt((double)v[i]) / bc2s + (double)eps;
        p[i] = (float)((double)p[i] - step_size * ((double)m[i] / denom));
    }
}

/* ---------------- whole-model forward / backward ---------------- */
typedef struct {
    int B, H;
    int hs[6];                 /* image spatial sizes per layer */
    int ts[5];                 /* sound time sizes per layer: 100,48,23,11,5 */
    float *img_act[6];         /* [0] = scaled f32 image */
    float *snd_act[2][5];      /* [pos|neg][layer]; [.][0] points at caller MFCC (copied) */
    float *ih[1], *sh[2];      /* hidden (B,128) post-ReLU */
    float *ie_raw, *se_raw[2]; /* pre-normalise (B,3) */
    float *ie_nrm, *se_nrm[2];
    float *ie, *se[2];         /* normalised */
} acts_t;

static acts_t *acts_new(int B, int H) {
    acts_t *A = (acts_t *)calloc(1, sizeof(acts_t));
    A->B = B; A->H = H;
    A->hs[0] = H;
    for (int i = 0; i < 5; i++) A->hs[i + 1] = out_sz(A->hs[i]);
    A->ts[0] = SND_T;
    A->ts[1] = (SND_T - 5) / 2 + 1;
    for (int i = 1; i < 4; i++) A->ts[i + 1] = (A->ts[i] - 3) / 2 + 1;
    for (int i = 0; i < 6; i++)
        A->img_act[i] = (float *)malloc(sizeof(float) * (size_t)B * IMG_CH[i] * A->hs[i] * A->hs[i]);
    for (int s = 0; s < 2; s++) {
        A->snd_act[s][0] = (float *)malloc(sizeof(float) * (size_t)B * SND_T * SND_F);
        for (int i = 1; i < 5; i++) A->snd_act[s][i] = (float *)malloc(sizeof(float) * (size_t)B * 32 * A->ts[i]);
        A->sh[s] = (float *)malloc(sizeof(float) * (size_t)B * HID);
        A->se_raw[s] = (float *)malloc(sizeof(float) * (size_t)B * EMB);
        A->se_nrm[s] = (float *)malloc(sizeof(float) * (size_t)B);
        A->se[s] = (float *)malloc(sizeof(float) * (size_t)B * EMB);
    }
    A->ih[0] = (float *)malloc(sizeof(float) * (size_t)B * HID);
    A->ie_raw = (float *)malloc(sizeof(float) * (size_t)B * EMB);
    A->ie_nrm = (float *)malloc(sizeof(float) * (size_t)B);
    A->ie = (float *)malloc(sizeof(float) * (size_t)B * EMB);
    return A;
}

static void acts_free(acts_t *A) {
    for (int i = 0; i < 6; i++) free(A->img_act[i]);
    for (int s = 0; s < 2; s++) {
        for (int i = 0; i < 5; i++) free(A->snd_act[s][i]);
        free(A->sh[s]); free(A->se_raw[s]); free(A->se_nrm[s]); free(A->se[s]);
    }
    free(A->ih[0]); free(A->ie_raw); free(A->ie_nrm); free(A->ie);
    free(A);
}

static void snd_dims(int layer, int *H, int *W) { /* input dims of sound layer */
    static const int T[5] = {100, 48, 23, 11, 5};
    *H = T[layer]; *W = layer == 0 ? SND_F : 1;
}

static void forward(const float *P, acts_t *A, const uint8_t *img_u8, const float *img_f32,
                    const float *pos, const float *neg) {
    layout_t L; make_layout(&L);
    int B = A->B, H = A->H;
    size_t n0 = (size_t)B * 3 * H * H;
    if (img_u8) for (size_t i = 0; i < n0; i++) A->img_act[0][i] = (float)img_u8[i] / 255.f;
    else memcpy(A->img_act[0], img_f32, n0 * sizeof(float));
    for (int i = 0; i < 5; i++)
        conv_fwd(A->img_act[i], P + L.img_w[i], P + L.img_b[i], A->img_act[i + 1],
                 B, IMG_CH[i], A->hs[i], A->hs[i], IMG_CH[i + 1], 3, 3, 2, 2, 1, 1);
    linear_fwd(A->img_act[5], P + L.ih_w0, P + L.ih_b0, A->ih[0], B, L.img_feat, HID, 1);
    linear_fwd(A->ih[0], P + L.ih_w1, P + L.ih_b1, A->ie_raw, B, HID, EMB, 0);
    orc_normalize_fwd(A->ie_raw, A->ie, A->ie_nrm, B);
    const float *snd[2] = {pos, neg};
    for (int s = 0; s < 2; s++) {
        if (!snd[s]) continue;
        memcpy(A->snd_act[s][0], snd[s], sizeof(float) * (size_t)B * SND_T * SND_F);
        for (int i = 0; i < 4; i++) {
            int h, w; snd_dims(i, &h, &w);
            conv_fwd(A->snd_act[s][i], P + L.snd_w[i], P + L.snd_b[i], A->snd_act[s][i + 1],
                     B, SND_CH[i], h, w, SND_CH[i + 1], SND_KH[i], SND_KW[i], 2, 1, 0, 0);
        }
        linear_fwd(A->snd_act[s][4], P + L.sh_w0, P + L.sh_b0, A->sh[s], B, L.snd_feat, HID, 1);
        linear_fwd(A->sh[s], P + L.sh_w1, P + L.sh_b1, A->se_raw[s], B, HID, EMB, 0);
        orc_normalize_fwd(A->se_raw[s], A->se[s], A->se_nrm[s], B);
    }
}

/* Encoder forward. Any of img_u8/img_f32 (exactly one), pos, neg given; outputs may be NULL. */
int orc_kuka_forward(const float *P, int B, int H, const uint8_t *img_u8, const float *img_f32,
                     const float *pos, const float *neg,
                     float *image_feat, float *pos_feat, float *neg_feat, float *image_raw, float *pos_raw) {
    acts_t *A = acts_new(B, H);
    forward(P, A, img_u8, img_f32, pos, neg);
    if (image_feat) memcpy(image_feat, A->ie, sizeof(float) * B * EMB);
    if (pos_feat && pos) memcpy(pos_feat, A->se[0], sizeof(float) * B * EMB);
    if (neg_feat && neg) memcpy(neg_feat, A->se[1], sizeof(float) * B * EMB);
    if (image_raw) memcpy(image_raw, A->img_act[5], sizeof(float) * (size_t)B * 576);
    if (pos_raw && pos) memcpy(pos_raw, A->snd_act[0][4], sizeof(float) * (size_t)B * 160);
    acts_free(A);
    return 0;
}

/* One training step's forward + loss + backward: G (same arena layout as P) receives
 * d(loss)/d(param).  Returns loss; also the three embeddings when non-NULL. */
float orc_kuka_loss_grad(const float *P, float *G, int B, int H, const uint8_t *img_u8, const float *img_f32,
                         const float *pos, const float *neg, float margin,
                         float *image_feat, float *pos_feat, float *neg_feat) {
    layout_t L; make_layout(&L);
    acts_t *A = acts_new(B, H);
    forward(P, A, img_u8, img_f32, pos, neg);
    memset(G, 0, sizeof(float) * L.total);
    float *ga = (float *)malloc(sizeof(float) * B * EMB), *gp = (float *)malloc(sizeof(float) * B * EMB),
          *gn = (float *)malloc(sizeof(float) * B * EMB);
    float loss = orc_triplet(A->ie, A->se[0], A->se[1], B, margin, ga, gp, gn);
    if (image_feat) memcpy(image_feat, A->ie, sizeof(float) * B * EMB);
    if (pos_feat) memcpy(pos_feat, A->se[0], sizeof(float) * B * EMB);
    if (neg_feat) memcpy(neg_feat, A->se[1], sizeof(float) * B * EMB);

    float *graw = (float *)malloc(sizeof(float) * B * EMB);
    float *ghid = (float *)malloc(sizeof(float) * (size_t)B * HID);
    /* image branch */
    {
        orc_normalize_bwd(A->ie, A->ie_nrm, ga, graw, B);
        linear_bwd(A->ih[0], P + L.ih_w1, A->ie_raw, graw, ghid, G + L.ih_w1, G + L.ih_b1, B, HID, EMB, 0);
        float *gfeat = (float *)malloc(sizeof(float) * (size_t)B * L.img_feat);
        linear_bwd(A->img_act[5], P + L.ih_w0, A->ih[0], ghid, gfeat, G + L.ih_w0, G + L.ih_b0, B, L.img_feat, HID, 1);
        float *gy = gfeat;
        for (int i = 4; i >= 0; i--) {
            float *gx = NULL;
            if (i > 0) gx = (float *)malloc(sizeof(float) * (size_t)B * IMG_CH[i] * A->hs[i] * A->hs[i]);
            conv_bwd(A->img_act[i], P + L.img_w[i], A->img_act[i + 1], gy, gx, G + L.img_w[i], G + L.img_b[i],
                     B, IMG_CH[i], A->hs[i], A->hs[i], IMG_CH[i + 1], 3, 3, 2, 2, 1, 1);
            free(gy);
            gy = gx;
        }
    }
    /* sound branch, positive then negative: shared weights, grads accumulate */
    float *gsnd[2] = {gp, gn};
    float *tmpw = (float *)malloc(sizeof(float) * 6400), *tmpb = (float *)malloc(sizeof(float) * 32);
    for (int s = 0; s < 2; s++) {
        orc_normalize_bwd(A->se[s], A->se_nrm[s], gsnd[s], graw, B);
        linear_bwd(A->sh[s], P + L.sh_w1, A->se_raw[s], graw, ghid, G + L.sh_w1, G + L.sh_b1, B, HID, EMB, 0);
        float *gfeat = (float *)malloc(sizeof(float) * (size_t)B * L.snd_feat);
        linear_bwd(A->snd_act[s][4], P + L.sh_w0, A->sh[s], ghid, gfeat, G + L.sh_w0, G + L.sh_b0, B, L.snd_feat, HID, 1);
        float *gy = gfeat;
        for (int i = 3; i >= 0; i--) {
            int h, w; snd_dims(i, &h, &w);
            float *gx = NULL;
            if (i > 0) gx = (float *)malloc(sizeof(float) * (size_t)B * SND_CH[i] * h * w);
            int nw = SND_CH[i + 1] * SND_CH[i] * SND_KH[i] * SND_KW[i];
            conv_bwd(A->snd_act[s][i], P + L.snd_w[i], A->snd_act[s][i + 1], gy, gx, tmpw, tmpb,
                     B, SND_CH[i], h, w, SND_CH[i + 1], SND_KH[i], SND_KW[i], 2, 1, 0, 0);
            for (int k = 0; k < nw; k++) G[L.snd_w[i] + k] += tmpw[k];
            for (int k = 0; k < SND_CH[i + 1]; k++) G[L.snd_b[i] + k] += tmpb[k];
            free(gy);
            gy = gx;
        }
    }
    free(tmpw); free(tmpb); free(graw); free(ghid); free(ga); free(gp); free(gn);
    acts_free(A);
    return loss;
}

/* Stand-alone layer entry points used by the HIP kernel unit tests. */
void orc_conv3x3s2_fwd(const float *x, const float *w, const float *b, float *y, int B, int Ci, int H, int Co) {
    conv_fwd(x, w, b, y, B, Ci, H, H, Co, 3, 3, 2, 2, 1, 1);
}
void orc_conv3x3s2_bwd(const float *x, const float *w, const float *y, float *dy, float *dx, float *dw, float *db,
                       int B, int Ci, int H, int Co) {
    conv_bwd(x, w, y, dy, dx, dw, db, B, Ci, H, H, Co, 3, 3, 2, 2, 1, 1);
}
