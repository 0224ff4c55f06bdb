"""GPU tests of the iTHOR model's bf16 mode (BASELINE config 4): the staged 11x5 sound convolution kernels of
csrc/snd_bf16.hip, layer by layer against a float64 convolution of the SAME bf16-rounded operands (so the only
difference left is the fp32 summation order: tolerance 2e-5 of the layer's scale)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(h):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 600, 40), representationDim=3)


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def sounds(B, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 1, 600, 40, generator=g) * 3).cuda(), (torch.randn(B, 1, 600, 40, generator=g) * 3).cuda()


@pytest.mark.parametrize("B", [1, 3])
def test_conv2_forward_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, 11 + B)
    with torch.no_grad():
        m(None, pos, neg)
    from var_amd._lib import Context
    ctx = Context.get(0)
    n = 2 * B
    s1 = ctx.debug_buffer("ithor_s1")[:n * 64 * 300 * 20].view(n, 64, 300, 20).cpu()
    s2 = ctx.debug_buffer("ithor_s2")[:n * 64 * 150 * 13].view(n, 64, 150, 13).cpu()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    wk = [k for k, v in sd.items() if tuple(v.shape) == (64, 64, 11, 5)][0]
    w, b = sd[wk], sd[wk.replace("weight", "bias")]
    ref = torch.relu(torch.nn.functional.conv2d(bf16_round(s1), bf16_round(w), b.double(), stride=2, padding=(5, 5)))
    assert ref.shape == s2.shape
    scale = float(ref.abs().max())
    assert scale > 0.1
    err = float((s2.double() - ref).abs().max())
    assert err < 2e-5 * max(scale, 1.0), (err, scale)
    assert float((s2 > 0).float().mean()) > 0.05                       # not trivially all-zero


def _after_backward(var_amd, B, seed):
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, seed)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed)).cuda()
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(img, pos, neg)
    from var_amd._lib import Context
    ctx = Context.get(0)
    n = 2 * B
    buf = {k: ctx.debug_buffer("ithor_" + k).cpu() for k in ("s1", "s2", "gs1", "gs2")}
    shapes = {"s1": (n, 64, 300, 20), "gs1": (n, 64, 300, 20), "s2": (n, 64, 150, 13), "gs2": (n, 64, 150, 13)}
    buf = {k: v[:int(np.prod(shapes[k]))].view(shapes[k]) for k, v in buf.items()}
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    wk = [k for k, v in sd.items() if tuple(v.shape) == (64, 64, 11, 5)][0]
    return m, tr, buf, sd[wk], wk


@pytest.mark.parametrize("B", [1, 2])
def test_conv2_data_gradient_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    m, tr, buf, w, _ = _after_backward(var_amd, B, 21 + B)
    gy = buf["gs2"]
    assert float(gy.abs().max()) > 0
    ref = torch.nn.functional.conv_transpose2d(bf16_round(gy), bf16_round(w), stride=2, padding=(5, 5), output_padding=(1, 1))
    ref = ref * (bf16_round(buf["s1"]) > 0)
    assert ref.shape == buf["gs1"].shape
    scale = float(ref.abs().max())
    err = float((buf["gs1"].double() - ref).abs().max())
    assert err < 2e-5 * scale, (err, scale)
    # conv 1's bias gradient comes out of the same kernel's store: the channel sums of what it wrote
    g, o = tr.grads.cpu().double(), 0
    for k, p in m.named_parameters():
        if k == "cnn.0.bias":
            got = g[o:o + p.numel()]
        o += p.numel()
    want = buf["gs1"].double().sum((0, 2, 3))
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max() + buf["gs1"].abs().double().sum((0, 2, 3)).max() * 1e-2)


@pytest.mark.parametrize("B", [1, 3])
def test_conv2_weight_gradient_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    m, tr, buf, w, wk = _after_backward(var_amd, B, 31 + B)
    g = tr.grads.cpu()
    o = 0
    for k, p in m.named_parameters():
        if k == wk:
            got = g[o:o + p.numel()].view(p.shape)
        o += p.numel()
    ref = torch.nn.grad.conv2d_weight(bf16_round(buf["s1"]), (64, 64, 11, 5), bf16_round(buf["gs2"]), stride=2, padding=(5, 5))
    scale = float(ref.abs().max())
    assert scale > 0
    err = float((got.double() - ref).abs().max())
    assert err < 1e-4 * scale, (err, scale)


def test_gru_recurrence_bf16_kernel_vs_float64_emulation(var_amd):
    """The fused step kernel (gru_bf16.hip): final hidden states of both directions against a float64 recurrence that
    rounds the same operands (x, W_ih, h, W_hh) to bf16 before each product and keeps everything else exact."""
    B = 2
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, 77)
    with torch.no_grad():
        out = m(None, pos, neg)
    from var_amd._lib import Context
    n = 2 * B
    x = Context.get(0).debug_buffer("ithor_s3")[:n * 73 * 448].view(n, 73, 448).cpu()
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    hs = []
    for sfx in ("", "_reverse"):
        w_ih, w_hh = sd["rnn.weight_ih_l0" + sfx], sd["rnn.weight_hh_l0" + sfx]
        b_ih, b_hh = sd["rnn.bias_ih_l0" + sfx], sd["rnn.bias_hh_l0" + sfx]
        gi = bf16_round(x) @ bf16_round(w_ih.float()).T + b_ih                      # (n, 73, 1536)
        h = torch.zeros(n, 512, dtype=torch.float64)
        order = range(73) if sfx == "" else range(72, -1, -1)
        for t in order:
            gh = bf16_round(h.float()) @ bf16_round(w_hh.float()).T + b_hh
            r = torch.sigmoid(gi[:, t, :512] + gh[:, :512])
            z = torch.sigmoid(gi[:, t, 512:1024] + gh[:, 512:1024])
            nn_ = torch.tanh(gi[:, t, 1024:] + r * gh[:, 1024:])
            h = (1 - z) * nn_ + z * h
        hs.append(h)
    ref = torch.cat(hs, dim=1)                                                      # (n, 1024): [pos clips | neg clips]
    got = out["pos_sound_raw"].cpu().double()
    err = float((got - ref[:B]).abs().max())
    assert err < 2e-4, err
    assert float(ref.abs().max()) > 0.05


def test_gru_and_sound_gradients_bf16_close_to_fp32(var_amd):
    """Backward of the fused GRU step kernel and the staged sound convolutions: per-tensor gradients of the sound
    branch against the fp32 path on the same batch (bf16 rounding of the operands only: a few per cent in L2)."""
    B = 3
    pos, neg = sounds(B, 91)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
    grads = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(5)
        m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision(prec)
        tr = var_amd.IthorTrainer(m)
        tr.loss_and_grads(img, pos, neg)
        g = tr.grads.cpu().double()
        o, d = 0, {}
        for k, p in m.named_parameters():
            d[k] = g[o:o + p.numel()]
            o += p.numel()
        grads[prec] = d
    worst = {}
    for k in grads["fp32"]:
        if k.startswith(("rnn.", "cnn.", "soundTriplet.")):
            a, b = grads["bf16"][k], grads["fp32"][k]
            worst[k] = float((a - b).norm() / (b.norm() + 1e-30))
    # measured: GRU 0.6-0.8 %, heads 0.5-2.8 %, convolutions 5-8 % (ReLU decisions flip on rounding differences)
    assert max(v for k, v in worst.items() if k.startswith("rnn.")) < 0.02, worst
    assert max(v for k, v in worst.items() if k.startswith("soundTriplet.")) < 0.06, worst
    assert max(v for k, v in worst.items() if k.startswith("cnn.")) < 0.15, worst


@pytest.mark.parametrize("akf,bkf", [(1, 1), (0, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("M,N,K,nsplit,add", [(128, 128, 32, 1, 0), (448, 200, 96, 1, 0), (1536, 292, 448, 1, 0),
                                              (448, 1536, 1472, 4, 0), (512, 1536, 2336, 8, 0), (132, 68, 64, 1, 1)])
def test_dense_bf16_kernel_vs_float64_on_rounded_operands(var_amd, akf, bkf, M, N, K, nsplit, add):
    """dense_bf16.h through var_debug_ithor_dense: all four operand layouts (K-contiguous -> ds_read_b128 image,
    index-contiguous -> transposed reads), ragged M / N, split-K slabs and the accumulate mode."""
    import ctypes
    from var_amd._lib import Context
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    ctx = Context.get(0)
    m._ensure_plan(ctx, 2)
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + akf * 2 + bkf)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    a_dev = (A if akf else A.T).contiguous().cuda()
    b_dev = (B.T if bkf else B).contiguous().cuda()
    ns = max(nsplit, 1)
    out = torch.full((ns, N, M), 0.5 if add else float("nan"), device="cuda")
    rc = ctx.lib.var_debug_ithor_dense(ctx.handle, None, akf, bkf, a_dev.data_ptr(), b_dev.data_ptr(), out.data_ptr(), M, N, K,
                                       nsplit, add)
    torch.cuda.synchronize()
    assert rc == 1, rc                                           # the staged kernel ran, not the fallback
    ref = (bf16_round(A) @ bf16_round(B)).T                      # (N, M): C[m + n*M]
    got = out.double().cpu()
    if nsplit > 1:
        got = got.sum(0)                                         # trimmed splits write zeros
    else:
        got = got[0] - (0.5 if add else 0.0)
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err < 2e-5 * float(ref.abs().max()) * max(1.0, (K / 512) ** 0.5), err


@pytest.mark.parametrize("B", [1, 2])
def test_conv3_forward_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    """The 7x3 stride-2 convolution through the geometry-templated staged kernel, reading conv 2's own C8 bf16 image
    (written by conv 2's store) and writing the GRU's (clip, 73, 448) sequence."""
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, 41 + B)
    with torch.no_grad():
        m(None, pos, neg)
    from var_amd._lib import Context
    ctx = Context.get(0)
    n = 2 * B
    s2 = ctx.debug_buffer("ithor_s2")[:n * 64 * 150 * 13].view(n, 64, 150, 13).cpu()
    s3 = ctx.debug_buffer("ithor_s3")[:n * 73 * 448].view(n, 73, 64, 7).permute(0, 2, 1, 3).cpu()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    wk = [k for k, v in sd.items() if tuple(v.shape) == (64, 64, 7, 3)][0]
    ref = torch.relu(torch.nn.functional.conv2d(bf16_round(s2), bf16_round(sd[wk]), sd[wk.replace("weight", "bias")].double(),
                                                stride=2, padding=(1, 1)))
    assert ref.shape == s3.shape
    scale = float(ref.abs().max())
    err = float((s3.double() - ref).abs().max())
    assert scale > 0.05 and err < 2e-5 * max(scale, 1.0), (err, scale)


@pytest.mark.parametrize("B", [1, 2])
def test_conv3_data_gradient_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    """conv 3's staged data-gradient kernel: gs2 against the transposed convolution of the bf16-rounded operands, masked by
    conv 2's sign words; its by-products: conv 2's bias gradient (channel sums of gs2) and -- checked through conv 2's own
    gradient tests above -- the C8 image of gs2."""
    m, tr, buf, _, _ = _after_backward(var_amd, B, 51 + B)
    from var_amd._lib import Context
    n = 2 * B
    gs3 = Context.get(0).debug_buffer("ithor_gs3")[:n * 73 * 448].view(n, 73, 64, 7).permute(0, 2, 1, 3).cpu()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    wk = [k for k, v in sd.items() if tuple(v.shape) == (64, 64, 7, 3)][0]
    ref = torch.nn.functional.conv_transpose2d(bf16_round(gs3), bf16_round(sd[wk]), stride=2, padding=(1, 1), output_padding=(1, 0))
    ref = ref * (bf16_round(buf["s2"]) > 0)
    assert ref.shape == buf["gs2"].shape
    scale = float(ref.abs().max())
    err = float((buf["gs2"].double() - ref).abs().max())
    assert scale > 0 and err < 2e-5 * scale, (err, scale)
    g, o = tr.grads.cpu().double(), 0
    for k, p in m.named_parameters():
        if k == wk.replace("weight", "bias").replace("cnn.4", "cnn.2"):
            got = g[o:o + p.numel()]
        o += p.numel()
    want = buf["gs2"].double().sum((0, 2, 3))
    assert float((got - want).abs().max()) < 1e-5 * float(buf["gs2"].abs().double().sum((0, 2, 3)).max())


@pytest.mark.parametrize("B", [1, 3])
def test_conv3_weight_gradient_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    m, tr, buf, _, _ = _after_backward(var_amd, B, 61 + B)
    from var_amd._lib import Context
    n = 2 * B
    gs3 = Context.get(0).debug_buffer("ithor_gs3")[:n * 73 * 448].view(n, 73, 64, 7).permute(0, 2, 1, 3).cpu()
    g, o = tr.grads.cpu(), 0
    for k, p in m.named_parameters():
        if tuple(p.shape) == (64, 64, 7, 3):
            got = g[o:o + p.numel()].view(p.shape)
        o += p.numel()
    ref = torch.nn.grad.conv2d_weight(bf16_round(buf["s2"]), (64, 64, 7, 3), bf16_round(gs3), stride=2, padding=(1, 1))
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max())
    assert scale > 0 and err < 1e-4 * scale, (err, scale)


@pytest.mark.parametrize("B", [1, 3])
def test_conv1_forward_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    """The 11x11 one-channel layer (taps as the MFMA k index, whole clip in LDS); its C8 image / sign words are what the
    conv-2 tests above run on."""
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, 71 + B)
    with torch.no_grad():
        m(None, pos, neg)
    from var_amd._lib import Context
    n = 2 * B
    s1 = Context.get(0).debug_buffer("ithor_s1")[:n * 64 * 300 * 20].view(n, 64, 300, 20).cpu()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = torch.cat([pos, neg]).cpu()
    ref = torch.relu(torch.nn.functional.conv2d(bf16_round(x), bf16_round(sd["cnn.0.weight"]), sd["cnn.0.bias"].double(),
                                                stride=2, padding=(5, 5)))
    assert ref.shape == s1.shape
    scale = float(ref.abs().max())
    err = float((s1.double() - ref).abs().max())
    assert scale > 0.1 and err < 2e-5 * max(scale, 1.0), (err, scale)


@pytest.mark.parametrize("B", [1, 3])
def test_conv1_weight_gradient_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, 81 + B)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(B)).cuda()
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(img, pos, neg)
    from var_amd._lib import Context
    n = 2 * B
    gs1 = Context.get(0).debug_buffer("ithor_gs1")[:n * 64 * 300 * 20].view(n, 64, 300, 20).cpu()
    g, o = tr.grads.cpu(), 0
    for k, p in m.named_parameters():
        if k == "cnn.0.weight":
            got = g[o:o + p.numel()].view(p.shape)
        o += p.numel()
    x = torch.cat([pos, neg]).cpu()
    ref = torch.nn.grad.conv2d_weight(bf16_round(x), (64, 1, 11, 11), bf16_round(gs1), stride=2, padding=(5, 5))
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max())
    assert scale > 0 and err < 1e-4 * scale, (err, scale)


def test_image_layers_2_to_5_bf16_kernels_vs_float64_on_rounded_operands(var_amd):
    """img_bf16.hip: the 3x3 stride-1 layers (32 -> 32 at 96x96, 32 -> 64 at 48x48, 64 -> 64 at 24x24 as whole-image tiles,
    64 -> 128 at 12x12 as two-image tiles: the odd batch leaves one half empty), forward and data gradient (one kernel with
    transposed / flipped filters), against float64 convolutions of the bf16-rounded operands."""
    B = 3
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16", keep_fp32_activations=True)
    pos, neg = sounds(B, 17)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).cuda()
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(img, pos, neg)
    from var_amd._lib import Context
    ctx = Context.get(0)

    def buf(name, c, hw):
        return ctx.debug_buffer("ithor_" + name)[:B * c * hw * hw].view(B, c, hw, hw).cpu()

    a1, a2, p2, a3 = buf("a1", 32, 96), buf("a2", 32, 96), buf("p2", 32, 48), buf("a3", 64, 48)
    ga1, ga2, gp2, ga3 = buf("ga1", 32, 96), buf("ga2", 32, 96), buf("gp2", 32, 48), buf("ga3", 64, 48)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    w2, b2, w3, b3 = sd["imgBranch.2.weight"], sd["imgBranch.2.bias"], sd["imgBranch.5.weight"], sd["imgBranch.5.bias"]
    F = torch.nn.functional

    def close(got, ref, what):
        scale = float(ref.abs().max())
        err = float((got.double() - ref).abs().max())
        assert scale > 0 and err < 2e-5 * max(scale, 1e-3 if scale < 1e-3 else scale), (what, err, scale)

    close(a2, torch.relu(F.conv2d(bf16_round(a1), bf16_round(w2), b2.double(), padding=1)), "a2")
    close(a3, torch.relu(F.conv2d(bf16_round(p2), bf16_round(w3), b3.double(), padding=1)), "a3")
    close(ga1, F.conv_transpose2d(bf16_round(ga2), bf16_round(w2), padding=1) * (a1 > 0), "ga1")
    close(gp2, F.conv_transpose2d(bf16_round(ga3), bf16_round(w3), padding=1), "gp2")
    p3, a4, p4, a5 = buf("p3", 64, 24), buf("a4", 64, 24), buf("p4", 64, 12), buf("a5", 128, 12)
    gp3, ga4, gp4, ga5 = buf("gp3", 64, 24), buf("ga4", 64, 24), buf("gp4", 64, 12), buf("ga5", 128, 12)
    w4, b4, w5, b5 = sd["imgBranch.8.weight"], sd["imgBranch.8.bias"], sd["imgBranch.11.weight"], sd["imgBranch.11.bias"]
    close(a4, torch.relu(F.conv2d(bf16_round(p3), bf16_round(w4), b4.double(), padding=1)), "a4")
    close(a5, torch.relu(F.conv2d(bf16_round(p4), bf16_round(w5), b5.double(), padding=1)), "a5")
    close(gp3, F.conv_transpose2d(bf16_round(ga4), bf16_round(w4), padding=1), "gp3")
    close(gp4, F.conv_transpose2d(bf16_round(ga5), bf16_round(w5), padding=1), "gp4")
    # weight gradients (pixels as the MFMA k index, transposed LDS reads)
    g, o, got = tr.grads.cpu(), 0, {}
    for k, p in m.named_parameters():
        got[k] = g[o:o + p.numel()].view(p.shape)
        o += p.numel()
    # conv 1's bias gradient = the channel sums of ga1, taken in layer 2's data-gradient store; conv 2's = those of ga2 (pool kernel)
    for key, gmap in (("imgBranch.0.bias", ga1), ("imgBranch.2.bias", ga2)):
        want = gmap.double().sum((0, 2, 3))
        assert float((got[key].double() - want).abs().max()) < 1e-5 * float(gmap.abs().double().sum((0, 2, 3)).max()), key
    x0 = (img.cpu().float() / 255.0)
    for key, xin, gout in (("imgBranch.0.weight", x0, ga1), ("imgBranch.2.weight", a1, ga2), ("imgBranch.5.weight", p2, ga3)):
        ref = torch.nn.grad.conv2d_weight(bf16_round(xin), tuple(sd[key].shape), bf16_round(gout), padding=1)
        scale = float(ref.abs().max())
        err = float((got[key].double() - ref).abs().max())
        assert scale > 0 and err < 1e-4 * scale, (key, err, scale)


@pytest.mark.parametrize("h,B", [(84, 3), (96, 5)])
def test_bf16_mode_other_image_size_and_odd_batch_close_to_fp32(var_amd, h, B):
    """Shapes the staged image kernels do not cover (84x84: 42 / 21 / 10 / 5 maps, odd ones among them) fall back to the
    gather-GEMM inside the bf16 mode; odd batches exercise the ragged tiles of every staged kernel.  Loss and embeddings
    against the fp32 path on the same batch."""
    pos, neg = sounds(B, 200 + h)
    img = torch.randint(0, 256, (B, 3, h, h), dtype=torch.uint8, generator=torch.Generator().manual_seed(h)).cuda()
    out = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(5)
        m = var_amd.IthorVARPretextNet(cfg(h)).to("cuda").set_precision(prec)
        tr = var_amd.IthorTrainer(m)
        loss, feats = tr.loss_and_grads(img, pos, neg, feats=True)
        out[prec] = (loss.item(), feats.cpu().numpy(), tr.grads.cpu().numpy().astype(np.float64))
    assert abs(out["bf16"][0] - out["fp32"][0]) < 2e-3
    np.testing.assert_allclose(out["bf16"][1], out["fp32"][1], atol=8e-3)
    a, b = out["bf16"][2], out["fp32"][2]
    assert np.isfinite(a).all() and float(np.linalg.norm(a - b) / np.linalg.norm(b)) < 0.2


def test_bf16_mode_single_sound_inputs(var_amd):
    """VAR_forward's routing in the bf16 mode: only the positive or only the negative sound given (the staged sound kernels
    then see one clip source and half the clips); each must equal the corresponding half of the two-sound call."""
    B = 3
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16")
    pos, neg = sounds(B, 303)
    with torch.no_grad():
        both = m(None, pos, neg)
        p_both, n_both = both["sound_feat_positive"].clone(), both["sound_feat_negative"].clone()
        only_p = m(None, pos, None)["sound_feat_positive"].clone()
        only_n = m(None, None, neg)["sound_feat_negative"].clone()
    np.testing.assert_allclose(only_p.cpu().numpy(), p_both.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(only_n.cpu().numpy(), n_both.cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("B", [1, 3, 40, 100])
def test_gru_one_launch_per_pass_equals_one_launch_per_step(var_amd, B):
    """The persistent GRU kernels (73 steps in one launch, the workgroups of a clip slice handing the state over through
    memory) do the per-step kernels' arithmetic in the same order: loss and gradients must be bit-identical, and no
    hand-off may have timed out.  B = 40 is two clip slices, the second ragged (80 clips = 64 + 16); B = 100 four slices
    (200 clips: the resident-panel input projection and the ring variants of the dense kernel run at that size too)."""
    import ctypes
    from var_amd._lib import Context
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16")
    pos, neg = sounds(B, 17)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).cuda()
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(img, pos, neg)                      # (plans the context)
    ctx = Context.get(0)
    out = {}
    try:
        for mode in (0, 1, 1):
            assert ctx.lib.var_ithor_set_gru_sequence(ctx.handle, mode) >= 0
            loss, _ = tr.loss_and_grads(img, pos, neg)
            torch.cuda.synchronize()
            out.setdefault(mode, []).append((float(loss), tr.grads.clone()))
    finally:
        ctx.lib.var_ithor_set_gru_sequence(ctx.handle, 1)
    word = ctypes.c_uint(123)
    assert ctx.lib.var_ithor_gru_status(ctx.handle, ctypes.byref(word)) == 0 and word.value == 0
    (l0, g0), = out[0]
    # the persistent backward sums the GRU's bias gradients on its way (per-slice register sums) instead of a second pass
    # over the gate gradients: those four vectors differ in summation order only, everything else is bit-identical
    spans, o = {}, 0
    for k, prm in m.named_parameters():
        spans[k] = (o, o + prm.numel())
        o += prm.numel()
    bias = [k for k in spans if k.startswith("rnn.bias")]
    assert len(bias) == 4
    for l1, g1 in out[1]:
        assert torch.isfinite(g1).all()
        assert l1 == l0
        a, b = g1.clone(), g0.clone()
        for k in bias:
            lo, hi = spans[k]
            scale = float(b[lo:hi].abs().max())
            assert float((a[lo:hi] - b[lo:hi]).abs().max()) <= 2e-6 * scale + 1e-12, k
            a[lo:hi] = 0
            b[lo:hi] = 0
        assert torch.equal(a, b), float((a - b).abs().max())
    assert torch.equal(out[1][0][1], out[1][1][1])        # and the persistent form is deterministic


def test_gru_persistent_launch_ends_when_its_grid_is_incomplete(var_amd):
    """Every wait of the persistent GRU kernels is bounded.  With one workgroup of each hand-off group missing (what the
    resident part of a grid sees when the rest cannot be scheduled) the launch must END, report it in the status word, and
    the step must be loud -- NaN loss and gradient -- rather than half updated; afterwards the context works again."""
    import time
    from var_amd._lib import Context
    B = 3
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16")
    pos, neg = sounds(B, 23)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(8)).cuda()
    tr = var_amd.IthorTrainer(m)
    loss, _ = tr.loss_and_grads(img, pos, neg)
    torch.cuda.synchronize()
    good_loss, good = float(loss), tr.grads.clone()
    assert m.gru_status() == 0
    ctx = Context.get(0)
    assert ctx.lib.var_debug_ithor_gru_drop_workgroup(ctx.handle) == 0
    t0 = time.time()
    loss, _ = tr.loss_and_grads(img, pos, neg)
    torch.cuda.synchronize()
    assert time.time() - t0 < 20.0                        # ended (the waits expire after ~0.3 s)
    assert m.gru_status() != 0
    assert not np.isfinite(float(loss)) and not bool(torch.isfinite(tr.grads).all())
    assert ctx.lib.var_ithor_set_gru_sequence(ctx.handle, 1) == 1          # clears the status word
    assert m.gru_status() == 0
    loss, _ = tr.loss_and_grads(img, pos, neg)
    torch.cuda.synchronize()
    assert float(loss) == good_loss and torch.equal(tr.grads, good)


@pytest.mark.parametrize("akf,bkf,M,N,K,add,want", [(1, 1, 1536, 1024 + 128 * 3, 448, 0, 2), (1, 1, 256, 2048, 448, 1, 2),
                                                     (1, 1, 1536, 292, 448, 0, 1), (0, 1, 448, 1168, 1536, 1, 1),
                                                     (0, 0, 512, 1536, 2336, 0, 1), (1, 0, 256, 520, 1632, 0, 1),
                                                     (0, 1, 448, 300, 1640, 0, 1)])
def test_dense_products_on_bf16_operand_copies(var_amd, akf, bkf, M, N, K, add, want):
    """The big products read bf16 COPIES of their operands (add bit 1 of var_debug_ithor_dense): the tile kernel's bf16
    staging paths, and the resident-panel kernel that the GRU input projection takes (K = 448, both operands k-fast, many
    n tiles: return code 2) -- several n tiles per workgroup, groups with unequal tile counts, bias-free accumulate mode;
    the register-ring variant of the tile kernel (a k-fast operand, >= 32 k-steps) with trip counts that are not multiples
    of the ring (51 and 52 steps, a K tail of 8) and ragged N."""
    from var_amd._lib import Context
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16")
    ctx = Context.get(0)
    m._ensure_plan(ctx, 2)
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    a_dev = (A if akf else A.T).contiguous().cuda()
    b_dev = (B.T if bkf else B).contiguous().cuda()
    out = torch.full((N, M), 0.25 if add else float("nan"), device="cuda")
    rc = ctx.lib.var_debug_ithor_dense(ctx.handle, None, akf, bkf, a_dev.data_ptr(), b_dev.data_ptr(), out.data_ptr(), M, N, K,
                                       1, 2 | add)
    torch.cuda.synchronize()
    assert rc == want, rc
    ref = (bf16_round(A) @ bf16_round(B)).T
    got = out.double().cpu() - (0.25 if add else 0.0)
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err < 2e-5 * float(ref.abs().max()) * max(1.0, (K / 512) ** 0.5), err


@pytest.mark.parametrize("h", [96, 84])
def test_image_layer_6_as_a_dense_layer_vs_float64_on_rounded_operands(var_amd, h):
    """bf16 mode, batch >= 64: the last image convolution (3x3 stride 2 on the 6x6 | 5x5 pooled map) runs as two dense
    products over the filter scattered into a (1152, 128 HP HP) matrix.  Forward (bias + ReLU) and data gradient against
    float64 convolutions of the bf16-rounded operands."""
    B = 64
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(h)).to("cuda").set_precision("bf16")
    pos, neg = sounds(B, 31)
    img = torch.randint(0, 256, (B, 3, h, h), dtype=torch.uint8, generator=torch.Generator().manual_seed(h + 1)).cuda()
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(img, pos, neg)
    from var_amd._lib import Context
    ctx = Context.get(0)
    hp = 6 if h == 96 else 5

    def buf(name, c, hw):
        return ctx.debug_buffer("ithor_" + name)[:B * c * hw * hw].view(B, c, hw, hw).cpu()

    p5, a6, gp5, ga6 = buf("p5", 128, hp), buf("a6", 128, 3), buf("gp5", 128, hp), buf("ga6", 128, 3)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    w6, b6 = sd["imgBranch.14.weight"], sd["imgBranch.14.bias"]
    assert tuple(w6.shape) == (128, 128, 3, 3)
    F = torch.nn.functional

    def close(got, ref, what):
        scale = float(ref.abs().max())
        err = float((got.double() - ref).abs().max())
        assert scale > 0 and err < 3e-5 * scale, (what, err, scale)

    close(a6, torch.relu(F.conv2d(bf16_round(p5), bf16_round(w6), b6.double(), stride=2, padding=1)), "a6")
    ref = F.conv_transpose2d(bf16_round(ga6), bf16_round(w6), stride=2, padding=1, output_padding=hp - 5)
    assert tuple(ref.shape) == tuple(gp5.shape)
    close(gp5, ref, "gp5")
