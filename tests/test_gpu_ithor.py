"""GPU parity tests of the iTHOR model: the HIP path through the C
ABI (var_ithor_*) against (1) the fixture the reference class produced (tests/golden/ithor_h96.npz) and (2) the
CPU oracle (oracle/torch_oracle.py:IthorNetCPU, itself pinned to that fixture) on seeded inputs.
Tolerances: north_star's 1e-3 (fp32) on embeddings and loss; gradients by per-tensor L2 error."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.torch_oracle import ithor_seeded  # noqa: E402  (checker only)


def cfg(h):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 600, 40), representationDim=3)


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


@pytest.fixture(scope="module")
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "ithor_h96.npz")))


def cuda(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def seeded_model(var_amd, seed, h=96):
    torch.manual_seed(seed)
    return var_amd.IthorVARPretextNet(cfg(h)).to("cuda")


def l2_rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def test_forward_dict_vs_reference_fixture(var_amd, fx):
    m = seeded_model(var_amd, int(fx["seed"]))
    for k, v in m.state_dict().items():                      # the seed reproduces the reference's weights
        f = v.cpu().numpy().reshape(-1).astype(np.float64)
        np.testing.assert_array_equal(np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]]), fx["check." + k], err_msg=k)
    with torch.no_grad():
        d = m(cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"]))
    assert d["image_BCE"] is None and d["sound_BCE"] is None
    for k in ("image_feat", "sound_feat_positive", "sound_feat_negative"):
        np.testing.assert_allclose(d[k].cpu().numpy(), fx[k], atol=1e-4, rtol=0, err_msg=k)
    for k in ("image_feat_raw", "pos_sound_raw"):
        assert l2_rel(d[k].cpu().numpy(), fx[k]) < 1e-4, k
        np.testing.assert_allclose(d[k].cpu().numpy(), fx[k], atol=1e-4, rtol=1e-3, err_msg=k)
    # f32 image input (image/255 done by the caller, dataset.py:67-68) gives the same result as u8
    with torch.no_grad():
        d2 = m((cuda(fx["image"]) / 255.).float(), None, None)
    assert d2["sound_feat_negative"] is None and d2["pos_sound_raw"] is None
    np.testing.assert_allclose(d2["image_feat"].cpu().numpy(), d["image_feat"].cpu().numpy(), atol=2e-6)


def test_loss_and_gradients_vs_reference_fixture(var_amd, fx):
    m = seeded_model(var_amd, int(fx["seed"]))
    tr = var_amd.IthorTrainer(m)
    loss, feats = tr.loss_and_grads(cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"]), feats=True)
    assert abs(loss.item() - float(fx["losses"][0])) < 1e-4
    np.testing.assert_allclose(feats.cpu().numpy()[:, 0:3], fx["image_feat"], atol=1e-4)
    np.testing.assert_allclose(feats.cpu().numpy()[:, 6:9], fx["sound_feat_negative"], atol=1e-4)
    g = tr.grads.cpu().numpy()
    stride = int(fx["stride"])
    o = 0
    for k, p in m.named_parameters():
        gt = g[o:o + p.numel()]
        o += p.numel()
        ref_s, ref_n = fx["gsamp." + k], float(fx["gnorm." + k])
        assert abs(np.linalg.norm(gt.astype(np.float64)) - ref_n) <= 2e-3 * ref_n + 1e-8, k
        assert np.linalg.norm(gt[::stride].astype(np.float64) - ref_s) <= 2e-3 * np.linalg.norm(ref_s) + 1e-3 * ref_n / np.sqrt(max(1, p.numel() // stride)) + 1e-9, k


def test_autograd_path_equals_fused_path(var_amd, fx):
    m = seeded_model(var_amd, int(fx["seed"]))
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"]))
    fused = tr.grads.clone()
    d = m(cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"]))
    loss = torch.nn.TripletMarginLoss(margin=1.0, p=2)(d["image_feat"], d["sound_feat_positive"], d["sound_feat_negative"])
    loss.backward()
    assert abs(loss.item() - float(fx["losses"][0])) < 1e-4
    auto = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert l2_rel(auto.cpu().numpy(), fused.cpu().numpy()) < 1e-4


def test_two_adam_steps_vs_reference_fixture(var_amd, fx):
    m = seeded_model(var_amd, int(fx["seed"]))
    tr = var_amd.IthorTrainer(m, lr=1e-4, weight_decay=1e-6)
    img, pos, neg = cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"])
    losses = [tr.step(img, pos, neg).item() for _ in range(2)]
    np.testing.assert_allclose(losses, fx["losses"], atol=2e-4)
    stride = int(fx["stride"])
    for k, v in m.state_dict().items():
        got = v.cpu().numpy().reshape(-1)[::stride]
        # Adam's first steps move every weight by ~lr whatever the gradient's size: compare the movement
        np.testing.assert_allclose(got, fx["adam2." + k], atol=4e-5, err_msg=k)


@pytest.mark.parametrize("h,b", [(84, 3), (96, 5)])
def test_full_gradients_vs_oracle(var_amd, h, b):
    torch.set_num_threads(8)
    ref = ithor_seeded(123)
    with torch.no_grad():                                     # spread the embeddings so some hinges are inactive
        for k, p in ref.named_parameters():
            if "Triplet" in k and k.endswith("weight"):
                p.mul_(3.0)
    m = var_amd.IthorVARPretextNet(cfg(h))
    m.load_state_dict(ref.state_dict())
    m = m.to("cuda")
    rng = np.random.default_rng(5 + h)
    img = rng.integers(0, 256, size=(b, 3, h, h), dtype=np.uint8)
    snd = (rng.standard_normal((2 * b, 1, 600, 40)) * 6.0).astype(np.float32)
    snd[0, :, 350:] = 0.0
    pos, neg = snd[:b], snd[b:]
    a, p_, n_ = ref((torch.from_numpy(img) / 255.).float(), torch.from_numpy(pos), torch.from_numpy(neg))
    loss = torch.nn.TripletMarginLoss(margin=1.0, p=2)(a, p_, n_)
    loss.backward()
    tr = var_amd.IthorTrainer(m)
    l, feats = tr.loss_and_grads(cuda(img), cuda(pos), cuda(neg), feats=True)
    assert abs(l.item() - loss.item()) < 1e-4
    np.testing.assert_allclose(feats.cpu().numpy(), torch.cat([a, p_, n_], 1).detach().numpy(), atol=1e-4)
    g = tr.grads.cpu().numpy()
    o = 0
    worst = 0.0
    for k, q in ref.named_parameters():
        gt = g[o:o + q.numel()]
        o += q.numel()
        e = l2_rel(gt, q.grad.numpy())
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
    print("worst per-tensor L2 error", worst)


def test_none_inputs_and_cached_goal_sound(var_amd, fx):
    m = seeded_model(var_amd, int(fx["seed"]))
    img, pos, neg = cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"])
    with torch.no_grad():
        d = m(img, pos, None)                                 # pretext.py:131,188
        assert d["sound_feat_negative"] is None
        np.testing.assert_allclose(d["sound_feat_positive"].cpu().numpy(), fx["sound_feat_positive"], atol=1e-4)
        d = m(img, torch.full_like(pos, float("inf")), None)  # pretext_base.py:29-32: cached goal embedding
        assert d["pos_sound_raw"] is None
        np.testing.assert_allclose(d["sound_feat_positive"].cpu().numpy(), fx["sound_feat_positive"], atol=1e-4)
        d = m(None, neg, None)
        assert d["image_feat"] is None and d["image_feat_raw"] is None
        np.testing.assert_allclose(d["sound_feat_positive"].cpu().numpy(), fx["sound_feat_negative"], atol=1e-4)
        d = m(None, None, neg)
        np.testing.assert_allclose(d["sound_feat_negative"].cpu().numpy(), fx["sound_feat_negative"], atol=1e-4)
        img4 = torch.cat([(img / 255.).float(), torch.ones(2, 1, 96, 96, device="cuda")], dim=1)
        d = m(img4, None, None)                               # image[:, :3] (pretext_base.py:22)
        np.testing.assert_allclose(d["image_feat"].cpu().numpy(), fx["image_feat"], atol=1e-4)


def test_rejects_cpu_tensors_and_bad_shapes(var_amd):
    with pytest.raises(var_amd.VarHipError):
        var_amd.IthorVARPretextNet(types.SimpleNamespace(img_dim=(3, 64, 64), sound_dim=(1, 600, 40), representationDim=3))
    m = seeded_model(var_amd, 1)
    with pytest.raises(var_amd.VarHipError):
        m(torch.zeros(1, 3, 96, 96), None, None)
    with pytest.raises(var_amd.VarHipError):
        m(None, torch.zeros(1, 1, 100, 40, device="cuda"), None)


def test_psf_mfcc_vs_oracle(var_amd):
    """var_mfcc_psf (python_speech_features branch, Envs/audioLoader.py:158-161 + :241-252) against the numpy
    restatement: 6 s clips, ragged lengths, a clip shorter than one window, the empty class."""
    from oracle import mfcc_np
    rng = np.random.default_rng(9)
    nmax = 96000
    lens = [96000, 50001, 16000, 400, 300, 0, 95999]
    pcm = np.zeros((len(lens), nmax), dtype=np.int16)
    for i, n in enumerate(lens):
        t = np.arange(n) / 16000.0
        f0 = rng.uniform(100, 4000)
        pcm[i, :n] = np.round(np.clip(3000 * rng.standard_normal(n) + 8000 * np.sin(2 * np.pi * f0 * t), -32767, 32767))
    out = var_amd.mfcc_psf(cuda(pcm), torch.tensor(lens, dtype=torch.int32), out_frames=600).cpu().numpy()
    assert out.shape == (len(lens), 1, 600, 40)
    for i, n in enumerate(lens):
        if n == 0:
            assert np.all(out[i] == 0)
            continue
        ref = mfcc_np.mfcc_psf(pcm[i, :n])
        ref = mfcc_np.process_sound_feat(ref, (1, 600, 40))
        T = min(600, 1 if n <= 400 else 1 + int(np.ceil((n - 400) / 160)))
        assert np.all(out[i, 0, T:] == 0)
        np.testing.assert_allclose(out[i], ref, atol=2e-3, rtol=1e-4, err_msg=f"clip {i} len {n}")
    # truncation to fewer frames than the clip has (audioLoader.py:245-246)
    out2 = var_amd.mfcc_psf(cuda(pcm[:1]), None, out_frames=100).cpu().numpy()
    np.testing.assert_allclose(out2[0], out[0][:, :100], atol=1e-6)


def test_shard_gradients_add_up_to_the_full_batch(var_amd, fx):
    """Data-parallel convention: shard gradients computed with inv_count = 1/B_global sum to
    the full-batch gradient -- what the ONE all-reduce of IthorTrainer.step relies on."""
    m = seeded_model(var_amd, int(fx["seed"]))
    tr = var_amd.IthorTrainer(m)
    img, pos, neg = cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"])
    tr.loss_and_grads(img, pos, neg)
    full = tr.gbuf.clone()
    acc = torch.zeros_like(full)
    for r in range(2):
        tr.loss_and_grads(img[r:r + 1], pos[r:r + 1], neg[r:r + 1], global_batch=2)
        acc += tr.gbuf
    assert abs(acc[-1].item() - full[-1].item()) < 1e-6
    assert l2_rel(acc[:-1].cpu().numpy(), full[:-1].cpu().numpy()) < 1e-5


def test_train_loop_checkpoint_and_projection(var_amd, fx, tmp_path):
    """train_representation (VAR/pretext_VAR.py:44-91) with the iTHOR model: MultiStepLR [20,30]
    (Envs/ai2thor/config.py:47-48), legacy .pt that the reference-shaped module loads, progress.csv;
    project_representation = pretext.py:147-203 without the plot."""
    from oracle.torch_oracle import IthorNetCPU
    m = seeded_model(var_amd, int(fx["seed"]))
    img, pos, neg = cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"])
    gt = torch.tensor([1, 3])
    batches = lambda: iter([(img, pos, neg, gt)])
    losses = var_amd.train_representation(m, batches, epochs=2, milestones=(20, 30), save_dir=str(tmp_path),
                                          save_interval=10, log=lambda *a: None)
    np.testing.assert_allclose(losses, fx["losses"], atol=2e-4)
    sd = torch.load(os.path.join(str(tmp_path), "1.pt"), map_location="cpu")
    ref = IthorNetCPU()
    ref.load_state_dict(sd)                                   # same 36 keys and shapes as the reference
    assert open(os.path.join(str(tmp_path), "progress.csv")).read().splitlines()[0] == "avg_loss"
    a, s, g = var_amd.project_representation(m, batches)
    assert a.shape == (2, 3) and s.shape == (2, 3) and list(g) == [1, 3]
    with torch.no_grad():
        ra, rp, _ = ref((torch.from_numpy(fx["image"]) / 255.).float(), torch.from_numpy(fx["sound_positive"]),
                        torch.from_numpy(fx["sound_negative"]))
    np.testing.assert_allclose(a, ra.numpy(), atol=1e-4)
    np.testing.assert_allclose(s, rp.numpy(), atol=1e-4)


def test_step_from_pcm_equals_explicit_front_end(var_amd, fx):
    """IthorTrainer.step_from_pcm = mfcc_psf (python_speech_features branch) + step; clips gathered from a pool."""
    pool = var_amd.SyntheticTripletPool(16, hw=96, task_num=4, clips_per_class=4, seed=3)
    idx, cp = pool.sample_indices(3)
    img, pcm, lens = pool.gather(idx, cp)
    losses = []
    for use_pcm in (True, False):
        m = seeded_model(var_amd, int(fx["seed"]))
        tr = var_amd.IthorTrainer(m)
        if use_pcm:
            losses.append(tr.step_from_pcm(img, pcm, lens).item())
        else:
            f = var_amd.mfcc_psf(pcm, lens, out_frames=600)
            assert torch.all(f[lens == 0] == 0)
            losses.append(tr.step(img, f[:3], f[3:]).item())
        flat = m.flat_parameters().clone()
    assert losses[0] == losses[1] and np.isfinite(losses[0])


def test_gradients_are_bitwise_reproducible(var_amd, fx):
    """No float atomics anywhere in the step: split-K partial sums go to slabs folded in fixed order."""
    m = seeded_model(var_amd, int(fx["seed"]))
    tr = var_amd.IthorTrainer(m)
    img, pos, neg = cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"])
    tr.loss_and_grads(img, pos, neg)
    first = tr.gbuf.clone()
    for _ in range(3):
        tr.loss_and_grads(img, pos, neg)
        assert torch.equal(tr.gbuf, first)


def test_large_ragged_batch_is_consistent_with_its_parts(var_amd, fx):
    """Size-independent properties at a bench-like size the CPU oracle cannot reach: per-sample independence of the
    forward (B = 301 vs its two parts) and additivity of the gradient over a partition of the batch."""
    m = seeded_model(var_amd, int(fx["seed"]))
    g = torch.Generator(device="cuda").manual_seed(5)
    B = 301
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device="cuda", generator=g)
    pos = torch.randn((B, 1, 600, 40), device="cuda", generator=g) * 6
    neg = torch.randn((B, 1, 600, 40), device="cuda", generator=g) * 6
    with torch.no_grad():
        full = m(img, pos, neg)
        a = {k: v.clone() for k, v in full.items() if v is not None}
        lo = m(img[:173], pos[:173], neg[:173])
        lo = {k: v.clone() for k, v in lo.items() if v is not None}
        hi = m(img[173:], pos[173:], neg[173:])
    for k in ("image_feat", "sound_feat_positive", "sound_feat_negative", "image_feat_raw", "pos_sound_raw"):
        both = torch.cat([lo[k], hi[k]])
        np.testing.assert_allclose(a[k].cpu().numpy(), both.cpu().numpy(), atol=2e-5, rtol=1e-4, err_msg=k)
    tr = var_amd.IthorTrainer(m)
    tr.loss_and_grads(img, pos, neg)
    whole = tr.gbuf.clone()
    acc = torch.zeros_like(whole)
    for sl in (slice(0, 173), slice(173, B)):
        tr.loss_and_grads(img[sl], pos[sl], neg[sl], global_batch=B)
        acc += tr.gbuf
    assert abs(acc[-1].item() - whole[-1].item()) < 1e-5
    assert l2_rel(acc[:-1].cpu().numpy(), whole[:-1].cpu().numpy()) < 1e-4


def test_bf16_operand_mode_stays_close_to_fp32(var_amd, fx):
    """BASELINE config 4's precision (bf16 operands, fp32 accumulate; opt-in via set_precision): the fp32 path is the
    parity path, this bounds how far the bf16 one drifts from it -- embeddings 5e-3, loss 1e-3, gradient 15 % in L2
    (ReLU / max-pool routing decisions flip on rounding differences), and a short training run follows the same
    loss curve."""
    img, pos, neg = cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"])
    out = {}
    for prec in ("fp32", "bf16"):
        m = seeded_model(var_amd, int(fx["seed"])).set_precision(prec)
        tr = var_amd.IthorTrainer(m)
        loss, feats = tr.loss_and_grads(img, pos, neg, feats=True)
        loss0, g = loss.item(), tr.grads.clone()
        losses = [tr.step(img, pos, neg).item() for _ in range(6)]
        out[prec] = (loss0, feats.cpu().numpy(), g.cpu().numpy(), losses)
    assert abs(out["bf16"][0] - out["fp32"][0]) < 1e-3
    assert abs(out["fp32"][0] - float(fx["losses"][0])) < 1e-4          # switching back and forth leaves fp32 exact
    np.testing.assert_allclose(out["bf16"][1], out["fp32"][1], atol=5e-3)
    assert l2_rel(out["bf16"][2], out["fp32"][2]) < 0.15
    assert not np.array_equal(out["bf16"][2], out["fp32"][2])            # the switch does something
    np.testing.assert_allclose(out["bf16"][3], out["fp32"][3], atol=3e-2)
    assert out["bf16"][3][-1] < out["bf16"][3][0]


def test_graph_replayed_step_equals_eager_steps(var_amd, fx):
    """IthorTrainer.capture_step: three replays over static buffers (refreshed in place) = three eager steps, bit for bit."""
    pool = var_amd.SyntheticTripletPool(24, hw=96, task_num=4, clips_per_class=4, seed=9)
    B = 3
    batches = [pool.gather(*pool.sample_indices(B)) for _ in range(3)]
    eager = seeded_model(var_amd, int(fx["seed"]))
    te = var_amd.IthorTrainer(eager)
    le = []
    for img, pcm, lens in batches:
        le.append(te.step_from_pcm(img, pcm, lens).item())
    pe = eager.flat_parameters().clone()
    rep = seeded_model(var_amd, int(fx["seed"]))
    tr = var_amd.IthorTrainer(rep)
    s_img, s_pcm, s_len = (t.clone() for t in batches[0])
    # capture_step runs one eager step on whatever the static buffers hold; give it the first batch and undo nothing:
    # the comparison starts from the state after that step
    replay = tr.capture_step(s_img, s_pcm, s_len)
    lr_ = [tr.loss.item()]
    for img, pcm, lens in batches[1:]:
        s_img.copy_(img); s_pcm.copy_(pcm); s_len.copy_(lens)
        lr_.append(replay().item())
    assert lr_ == le
    assert torch.equal(rep.flat_parameters(), pe)
    assert tr.step_count == 3


def test_c_abi_error_paths(var_amd):
    """Errors come back as codes + var_last_error text, never as crashes: calls before a plan, NULL arguments, an odd
    PCM stride, a batch above the plan."""
    import ctypes
    from var_amd._lib import Context
    c = Context.get(0)
    lib = c.lib
    x = torch.zeros(16, device="cuda")
    assert lib.var_ithor_plan(c.handle, 0, 96) < 0 and b"batch" in lib.var_last_error(c.handle)
    assert lib.var_ithor_plan(c.handle, 2, 64) < 0 and b"3x3" in lib.var_last_error(c.handle)
    assert lib.var_ithor_plan(c.handle, 2, 96) == 0
    rc = lib.var_ithor_loss_grad(c.handle, None, x.data_ptr(), None, 1, 0, None, None, 2, 96, 1.0, 0.5, x.data_ptr(), None, None)
    assert rc < 0 and b"required" in lib.var_last_error(c.handle)
    rc = lib.var_ithor_encoder_fwd(c.handle, None, x.data_ptr(), None, 0, 0, None, None, 100000, 96, None, None, None, None, None, 0)
    assert rc < 0 and b"var_ithor_plan" in lib.var_last_error(c.handle)          # batch above any plan: refused before any launch
    rc = lib.var_mfcc_psf(c.handle, None, x.data_ptr(), x.data_ptr(), None, 1, 15, 10, x.data_ptr())
    assert rc < 0 and b"even" in lib.var_last_error(c.handle)
    rc = lib.var_inbatch_loss_fwd_bwd(c.handle, None, x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 1, ctypes.c_float(0.0),
                                      ctypes.c_float(1.0), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr())
    assert rc < 0
    rc = lib.var_armnet_forward(c.handle, None, x.data_ptr(), x.data_ptr(), 0, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                x.data_ptr(), x.data_ptr(), 100000, x.data_ptr(), x.data_ptr(), None, x.data_ptr())
    assert rc < 0 and b"var_armnet_plan" in lib.var_last_error(c.handle)
