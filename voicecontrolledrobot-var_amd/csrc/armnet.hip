// RL actor-critic forward on gfx950 (BASELINE config 5 second half):
// models/RL/arm_RL_model.py:7-134 `armNet_VAR` (96x96 branch: 8 convolutions / 3 max pools -> 1152, the motor /
// image / sound MLPs, one GRU(128 -> 512) step through NNBase._forward_gru's acting path, models/ppo/model.py:116-121,
// fusion and the actor / critic trunks) followed by DiagGaussian's mean layer (models/ppo/distributions.py:65-84), i.e.
// everything of Policy.act up to the sampling.  Inference only (the PPO update stays in PyTorch); every product is
// an instance of the gather-GEMM of gg.h, parameters are used in place in their state_dict() layouts.
#include <string.h>

#include "gg.h"

namespace {
constexpr int kCh[9] = {3, 32, 32, 64, 64, 128, 128, 256, 128};
constexpr int kRepr = 3, kRobot = 2, kRin = 128, kRh = 512, kAct = 128, kActions = 2, kFlat = 1152;

struct Lin { int w, b, in, out; };
struct ArmLayout {
    int g_wih, g_whh, g_bih, g_bhh;
    int cw[8], cb[8];
    Lin motor[3], cnn[2], im[2], im2, snd[3], fus[2], all[2], actor[2], critic[2], clin, mean;
    int logstd;
    int total;
};

ArmLayout make_layout() {
    ArmLayout L{};
    int o = 0;
    L.g_wih = o; o += 3 * kRh * kRin; L.g_whh = o; o += 3 * kRh * kRh; L.g_bih = o; o += 3 * kRh; L.g_bhh = o; o += 3 * kRh;
    for (int i = 0; i < 8; i++) { L.cw[i] = o; o += kCh[i + 1] * kCh[i] * 9; L.cb[i] = o; o += kCh[i + 1]; }
    auto lin = [&](int in, int out) { Lin l{o, o + in * out, in, out}; o += in * out + out; return l; };
    L.motor[0] = lin(kRepr + kRobot, 256); L.motor[1] = lin(256, 512); L.motor[2] = lin(512, 256);
    L.cnn[0] = lin(kFlat, 512); L.cnn[1] = lin(512, 256);
    L.im[0] = lin(256, 256); L.im[1] = lin(256, kRin);
    L.im2 = lin(kRh, 256);
    L.snd[0] = lin(kRepr, 128); L.snd[1] = lin(128, 256); L.snd[2] = lin(256, 256);
    L.fus[0] = lin(256, 512); L.fus[1] = lin(512, 256);
    L.all[0] = lin(256, 256); L.all[1] = lin(256, 128);
    L.actor[0] = lin(128, 128); L.actor[1] = lin(128, kAct);
    L.critic[0] = lin(128, 128); L.critic[1] = lin(128, 128);
    L.clin = lin(128, 1);
    L.mean = lin(kAct, kActions);
    L.logstd = o; o += kActions;
    L.total = o;
    return L;
}

struct arm_state {
    ArmLayout L;
    int maxB = 0;
    float* ws = nullptr;
    float *a[9] = {nullptr}, *p[4] = {nullptr};      // conv outputs 1..8, pooled maps 1..3
    float *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;   // (B,512) scratch rows
    float* slab = nullptr;
    float *flat_img = nullptr, *motor = nullptr, *sound = nullptr, *fusion = nullptr, *h0 = nullptr, *gi = nullptr, *gh = nullptr;
};

static __global__ void an_pool_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int H, int HP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = (int)(i % HP), py = (int)((i / HP) % HP);
    const long plane = i / ((long)HP * HP);
    const float* q = x + plane * H * H + (long)(2 * py) * H + 2 * px;
    y[i] = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[H], q[H + 1]));
}
// out = a + b (fusion sums), or out[b][:] = [u[b][:nu] | v[b][:nv]] (the motor input), or h * mask per row
static __global__ void an_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
static __global__ void an_cat_kernel(const float* __restrict__ u, int nu, const float* __restrict__ v, int nv, float* __restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (nu + nv)) return;
    const int b = i / (nu + nv), j = i - b * (nu + nv);
    out[i] = j < nu ? u[b * nu + j] : v[b * nv + j - nu];
}
static __global__ void an_mask_kernel(const float* __restrict__ h, const float* __restrict__ mask, float* __restrict__ out, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * H) out[i] = h[i] * mask[i / H];
}
// torch.nn.GRU cell (gate order r, z, n); gi / gh include their biases
static __global__ void an_gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                                          float* __restrict__ out, float* __restrict__ out2, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* a = gi + (long)b * 3 * H;
    const float* g = gh + (long)b * 3 * H;
    const float r = 1.f / (1.f + expf(-(a[j] + g[j])));
    const float z = 1.f / (1.f + expf(-(a[H + j] + g[H + j])));
    const float n = tanhf(a[2 * H + j] + r * g[2 * H + j]);
    const float v = (1.f - z) * n + z * h[i];
    out[i] = v;
    if (out2) out2[i] = v;
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }
#define AN_CHECK(c) VAR_HIP_CHECK(c, hipGetLastError())
#define RUN(x) do { int r_ = (x); if (r_ != VAR_OK) return r_; } while (0)

constexpr long kSlab = 8L << 20;             // floats of split-K scratch

template <class G, bool U8>
int conv(var_ctx* c, hipStream_t s, arm_state* st, const ConvDims& d, const void* x, const float* w, const float* bias, float* y) {
    ConvFwdP<G, U8, false> p{};
    p.M = d.B * d.HO * d.WO; p.N = d.COUT; p.K = d.CIN * G::KHW;
    const long out = (long)p.M * p.N;
    p.nsplit = gg_small_split(((p.M + GG_MT - 1) / GG_MT) * ((p.N + 63) / 64), p.K, out, kSlab);
    p.d = d; p.x = x; p.w = w; p.bias = bias; p.y = y; p.slab = st->slab; p.sstride = out;
    RUN(gg_launch(c, s, p));
    if (p.nsplit > 1) {
        hipLaunchKernelGGL(gg_finish_kernel, g1(out), dim3(256), 0, s, y, st->slab, out, p.nsplit, out, bias, d.COUT, d.HO * d.WO, 1);
        AN_CHECK(c);
    }
    return VAR_OK;
}
int linear(var_ctx* c, hipStream_t s, arm_state* st, const float* P, const Lin& l, const float* X, float* Y, int rows, int relu) {
    const long out = (long)rows * l.out;
    const int ns = gg_small_split(((l.out + GG_MT - 1) / GG_MT) * ((rows + 63) / 64), l.in, out, kSlab);
    if (ns > 1) {
        DenseP<true, true, 2> p{};
        p.M = l.out; p.N = rows; p.K = l.in; p.nsplit = ns;
        p.A = P + l.w; p.sam = l.in; p.sak = 1; p.Bm = X; p.sbk = 1; p.sbn = l.in; p.C = st->slab; p.scm = 1; p.scn = l.out; p.sC = out;
        RUN(gg_launch(c, s, p));
        hipLaunchKernelGGL(gg_finish_kernel, g1(out), dim3(256), 0, s, Y, st->slab, out, ns, out, P + l.b, l.out, 1, relu);
        AN_CHECK(c);
        return VAR_OK;
    }
    DenseP<true, true, 0> p{};
    p.M = l.out; p.N = rows; p.K = l.in; p.nsplit = 1;
    p.A = P + l.w; p.sam = l.in; p.sak = 1; p.Bm = X; p.sbk = 1; p.sbn = l.in; p.C = Y; p.scm = 1; p.scn = l.out;
    p.bias = P + l.b; p.relu = relu;
    return gg_launch(c, s, p);
}
}  // namespace

void armnet_free(var_ctx* c) {
    arm_state* st = (arm_state*)c->arm;
    if (!st) return;
    if (st->ws) (void)hipFree(st->ws);
    delete st;
    c->arm = nullptr;
}

extern "C" {

int var_armnet_param_count(void) { return make_layout().total; }

int var_armnet_plan(var_ctx* c, int max_batch) {
    if (!c) return VAR_ERR_ARG;
    if (max_batch < 1 || max_batch > 4096) { VAR_SET_ERR(c, "var_armnet_plan: batch %d outside 1..4096", max_batch); return VAR_ERR_ARG; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    arm_state* st = (arm_state*)c->arm;
    if (st && st->maxB >= max_batch) return VAR_OK;
    if (st) {      // retire (do not free) the superseded workspace: a captured act() graph may still replay on it
        if (st->ws) { int rc = retire_block(c, st->ws); if (rc != VAR_OK) return rc; }
        delete st;
        c->arm = nullptr;
    }
    c->plan_gen++;
    st = new arm_state();
    c->arm = st;
    st->L = make_layout();
    st->maxB = max_batch;
    const long B = max_batch;
    const int side[9] = {96, 96, 96, 48, 48, 24, 24, 5, 3};          // output side of conv l
    long total = 0;
    auto take = [&](long n) { long o = total; total += (n + 63) & ~63L; return o; };
    long oa[9], op[4];
    for (int l = 1; l <= 8; ++l) oa[l] = take(B * kCh[l] * side[l] * side[l]);
    op[1] = take(B * 32 * 48 * 48); op[2] = take(B * 64 * 24 * 24); op[3] = take(B * 128 * 12 * 12);
    const long ot0 = take(B * 512), ot1 = take(B * 512), ot2 = take(B * 512), ot3 = take(B * 512);
    const long ofl = take(B * 256), omo = take(B * 256), osn = take(B * 256), ofu = take(B * 256), oh0 = take(B * kRh);
    const long ogi = take(B * 3 * kRh), ogh = take(B * 3 * kRh), oslab = take(kSlab);
    VAR_HIP_CHECK(c, hipMalloc((void**)&st->ws, (size_t)total * sizeof(float)));
    float* w = st->ws;
    for (int l = 1; l <= 8; ++l) st->a[l] = w + oa[l];
    for (int l = 1; l <= 3; ++l) st->p[l] = w + op[l];
    st->t0 = w + ot0; st->t1 = w + ot1; st->t2 = w + ot2; st->t3 = w + ot3;
    st->flat_img = w + ofl; st->motor = w + omo; st->sound = w + osn; st->fusion = w + ofu; st->h0 = w + oh0;
    st->gi = w + ogi; st->gh = w + ogh; st->slab = w + oslab;
    return VAR_OK;
}

int var_armnet_forward(var_ctx* c, void* stream, const float* params, const void* image, int image_is_u8, long image_bstride,
                       const float* image_feat, const float* robot_pose, const float* goal_sound_feat,
                       const float* rnn_hxs, const float* masks, int B,
                       float* value, float* actor_features, float* action_mean, float* rnn_hxs_out) {
    if (!c) return VAR_ERR_ARG;
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    arm_state* st = (arm_state*)c->arm;
    if (!st || B > st->maxB) { VAR_SET_ERR(c, "var_armnet_forward: var_armnet_plan(%d) first", B); return VAR_ERR_PLAN; }
    if (!params || !image || !image_feat || !robot_pose || !goal_sound_feat || !rnn_hxs || !masks || !value ||
        !actor_features || !rnn_hxs_out || B < 1) {
        VAR_SET_ERR(c, "var_armnet_forward: NULL argument");
        return VAR_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const ArmLayout& L = st->L;
    const float* P = params;
    using S1 = Geo<3, 3, 1, 1, 1, 1>;
    using S2P0 = Geo<3, 3, 2, 2, 0, 0>;
    using S1P0 = Geo<3, 3, 1, 1, 0, 0>;
    auto dims = [&](int l, int hin, int stride, int pad) {
        return conv_dims(B, kCh[l - 1], hin, hin, kCh[l], 3, 3, stride, stride, pad, pad);
    };
    auto pool = [&](const float* x, float* y, int ch, int hin) -> int {
        const long n = (long)B * ch * (hin / 2) * (hin / 2);
        hipLaunchKernelGGL(an_pool_kernel, g1(n), dim3(256), 0, s, x, y, n, hin, hin / 2);
        AN_CHECK(c);
        return VAR_OK;
    };
    // imgCNN
    {
        ConvDims d = dims(1, 96, 1, 1);
        d.xb = image_bstride;
        if (image_is_u8) RUN((conv<S1, true>(c, s, st, d, image, P + L.cw[0], P + L.cb[0], st->a[1])));
        else RUN((conv<S1, false>(c, s, st, d, image, P + L.cw[0], P + L.cb[0], st->a[1])));
    }
    RUN((conv<S1, false>(c, s, st, dims(2, 96, 1, 1), st->a[1], P + L.cw[1], P + L.cb[1], st->a[2])));
    RUN(pool(st->a[2], st->p[1], 32, 96));
    RUN((conv<S1, false>(c, s, st, dims(3, 48, 1, 1), st->p[1], P + L.cw[2], P + L.cb[2], st->a[3])));
    RUN((conv<S1, false>(c, s, st, dims(4, 48, 1, 1), st->a[3], P + L.cw[3], P + L.cb[3], st->a[4])));
    RUN(pool(st->a[4], st->p[2], 64, 48));
    RUN((conv<S1, false>(c, s, st, dims(5, 24, 1, 1), st->p[2], P + L.cw[4], P + L.cb[4], st->a[5])));
    RUN((conv<S1, false>(c, s, st, dims(6, 24, 1, 1), st->a[5], P + L.cw[5], P + L.cb[5], st->a[6])));
    RUN(pool(st->a[6], st->p[3], 128, 24));
    RUN((conv<S2P0, false>(c, s, st, dims(7, 12, 2, 0), st->p[3], P + L.cw[6], P + L.cb[6], st->a[7])));
    RUN((conv<S1P0, false>(c, s, st, dims(8, 5, 1, 0), st->a[7], P + L.cw[7], P + L.cb[7], st->a[8])));
    // image_flatten = cnnMlp(flatten)
    RUN(linear(c, s, st, P, L.cnn[0], st->a[8], st->t0, B, 1));
    RUN(linear(c, s, st, P, L.cnn[1], st->t0, st->flat_img, B, 1));
    // motor = motorMlp(cat(image_feat, robot_pose))
    hipLaunchKernelGGL(an_cat_kernel, g1(B * 5), dim3(256), 0, s, image_feat, kRepr, robot_pose, kRobot, st->t0, B);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.motor[0], st->t0, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.motor[1], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.motor[2], st->t2, st->motor, B, 1));
    // imageMotor = imgMotorMlp(image_flatten + motor)
    hipLaunchKernelGGL(an_add_kernel, g1(B * 256), dim3(256), 0, s, st->flat_img, st->motor, st->t0, B * 256);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.im[0], st->t0, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.im[1], st->t1, st->t2, B, 1));                       // (B,128)
    // one GRU step from hxs * masks (models/ppo/model.py:118-121)
    hipLaunchKernelGGL(an_mask_kernel, g1(B * kRh), dim3(256), 0, s, rnn_hxs, masks, st->h0, B, kRh);
    AN_CHECK(c);
    {
        const Lin ih{L.g_wih, L.g_bih, kRin, 3 * kRh}, hh{L.g_whh, L.g_bhh, kRh, 3 * kRh};
        RUN(linear(c, s, st, P, ih, st->t2, st->gi, B, 0));
        RUN(linear(c, s, st, P, hh, st->h0, st->gh, B, 0));
        hipLaunchKernelGGL(an_gru_cell_kernel, g1(B * kRh), dim3(256), 0, s, st->gi, st->gh, st->h0, st->t3, rnn_hxs_out, B, kRh);
        AN_CHECK(c);
    }
    RUN(linear(c, s, st, P, L.im2, st->t3, st->t0, B, 1));                         // imageMotorRnn (B,256)
    // sound, fusion
    RUN(linear(c, s, st, P, L.snd[0], goal_sound_feat, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.snd[1], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.snd[2], st->t2, st->sound, B, 1));
    hipLaunchKernelGGL(an_add_kernel, g1(B * 256), dim3(256), 0, s, st->sound, st->flat_img, st->t1, B * 256);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.fus[0], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.fus[1], st->t2, st->fusion, B, 1));
    hipLaunchKernelGGL(an_add_kernel, g1(B * 256), dim3(256), 0, s, st->fusion, st->t0, st->t1, B * 256);
    AN_CHECK(c);
    RUN(linear(c, s, st, P, L.all[0], st->t1, st->t2, B, 1));
    RUN(linear(c, s, st, P, L.all[1], st->t2, st->t3, B, 1));                      // x (B,128)
    RUN(linear(c, s, st, P, L.critic[0], st->t3, st->t0, B, 1));
    RUN(linear(c, s, st, P, L.critic[1], st->t0, st->t1, B, 1));
    RUN(linear(c, s, st, P, L.clin, st->t1, value, B, 0));
    RUN(linear(c, s, st, P, L.actor[0], st->t3, st->t0, B, 1));
    RUN(linear(c, s, st, P, L.actor[1], st->t0, actor_features, B, 1));
    if (action_mean) RUN(linear(c, s, st, P, L.mean, actor_features, action_mean, B, 0));
    return VAR_OK;
}

}  // extern "C"
