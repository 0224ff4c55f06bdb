"""GPU parity tests added in round 2 (-m gpu, through the C ABI):
  * BASELINE configs[1] at the size it names: batch 256, 84x84, full-batch loss / embeddings / gradient arena against
    the torch restatement of the reference step, and a 10-step trajectory of the replayed HIP graph;
  * where a gradient tensor misses the tight tolerance, the miss is TRACED to ReLU units whose pre-activation lies
    within rounding of zero (the claim earlier tests only made in a comment);
  * the reference's ragged epoch (300 triplets, batch 128: 128 / 128 / 44) through the replayed step;
  * per-model packed weights (two models interleaved on one device), the saved-forward generation guard, gradient
    accumulation through autograd, and graphs that survive a re-plan of the workspace;
  * the in-batch-negatives head at configs[2]'s size (256 anchors x 4096 candidates)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import var_oracle as orc  # noqa: E402  (checker only)
from oracle import mfcc_np  # noqa: E402
from oracle.torch_oracle import CPUTrainer, KukaNetCPU  # noqa: E402


def cfg(h=84):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 100, 40), representationDim=3)


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def make_model(var_amd, sd, h=84):
    m = var_amd.VARPretextNet(cfg(h))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to("cuda")


def cuda(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def cpu_features(pool, clip_idx, lens):
    """MFCC features of pool clips by the numpy oracle (float32 output)."""
    out = np.zeros((len(clip_idx), 1, 100, 40), np.float32)
    clips = pool.clips.cpu().numpy()
    for i, (c, n) in enumerate(zip(clip_idx.tolist(), lens.tolist())):
        if n > 0:
            out[i] = mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(clips[c, :n]).astype(np.float32))
    return torch.from_numpy(out)


from tests._gpu_helpers import assert_grads_match_or_traced, torch_loss_grad  # noqa: E402


def test_config2_batch256_full_batch_parity(var_amd, golden_dir):
    """BASELINE configs[1]: batch 256, 84x84, fp32 -- loss, the three embeddings and all 213 478 gradients of the FULL
    batch against the CPU restatement of the reference step (<= 1e-3)."""
    sd = load(golden_dir, "kuka_weights2.npz")
    B = 256
    pool = var_amd.SyntheticTripletPool(512, hw=84, seed=3, clips_per_class=8).freeze_pairs()
    row = pool.epoch_index_table(B)[0]
    img = pool.images[row[:B].long()].contiguous()
    feats = var_amd.mfcc(pool.clips, row[3 * B:], out_frames=100, clip_index=row[B:3 * B])
    pos, neg = feats[:B].contiguous(), feats[B:].contiguous()
    m = make_model(var_amd, sd)
    tr = var_amd.VARTrainer(m)
    c = tr.ctx
    from var_amd._lib import ptr
    out = torch.empty((3, B, 3), device="cuda")                  # feats_out: three (B,3) blocks [image | pos | neg]
    c.ensure_plan(B, 84)
    tr._bind()
    c.check(c.lib.var_arm_loss_grad(c.handle, c.stream(), ptr(m.flat_parameters()), ptr(img), 1, img.stride(0), ptr(pos),
                                    ptr(neg), B, 84, 1.0, 1.0 / B, ptr(tr.gbuf), tr._loss_ptr(), ptr(out)), "loss_grad")
    net, l_ref, g_ref, (a, p, n), image_f32 = torch_loss_grad(sd, img.cpu(), pos.cpu(), neg.cpu())
    assert abs(tr.loss.item() - l_ref) < 1e-5
    emb = out.cpu().numpy()
    for got, ref, nm in ((emb[0], a, "image"), (emb[1], p, "pos"), (emb[2], n, "neg")):
        assert np.max(np.abs(got - ref)) < 1e-5, nm                 # north_star asks 1e-3; fp32 gives 1e-6
    # the fused training path (no embedding output) computes the same step
    tr.loss_and_grads(img, pos, neg)
    assert abs(tr.loss.item() - l_ref) < 1e-5
    worst, flips = assert_grads_match_or_traced(var_amd, tr, net, g_ref, image_f32, pos.cpu(), neg.cpu(), B)
    print("worst per-tensor gradient error", max(worst.values()), "ReLU gate flips", len(flips))


def test_config2_ten_step_trajectory_of_the_replayed_graph(var_amd, golden_dir):
    """10 optimisation steps at batch 256 on a fixed pool, HIP replayed graph (MFCC inside the step) against the CPU
    restatement fed the oracle's MFCC of the same clips: per-step loss within 1e-3 and, after step 10, the embeddings of
    a held-out batch within 1e-3 (SURVEY 8d config 2 'after K (e.g. 10) steps')."""
    sd = load(golden_dir, "kuka_weights.npz")
    B, steps = 256, 10
    pool = var_amd.SyntheticTripletPool(768, hw=84, seed=21, clips_per_class=3).freeze_pairs()
    table = pool.index_table(B, steps, drop_last=True)[:steps].contiguous()
    m = make_model(var_amd, sd)
    tr = var_amd.VARTrainer(m, lr=1e-4, weight_decay=1e-6)
    replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
    ref = CPUTrainer(state_dict=sd, lr=1e-4, weight_decay=1e-6)
    feat_cache = {}

    def feats_of(idx, lens):
        key = (tuple(idx.tolist()), tuple(lens.tolist()))
        if key not in feat_cache:
            feat_cache[key] = cpu_features(pool, idx, lens)
        return feat_cache[key]
    tcpu = table.cpu()
    for s in range(steps):
        r = tcpu[s]
        want = ref.step(pool.images[r[:B].long()].cpu(), feats_of(r[B:2 * B], r[3 * B:4 * B]), feats_of(r[2 * B:3 * B], r[4 * B:]))
        got = float(replay().item())
        assert abs(got - want) < 1e-3, (s, got, want)
        assert abs(got - want) < 2e-5 * (s + 1), (s, got, want)       # what fp32 actually delivers
    # held-out batch: items the 10 steps may or may not have seen, different clip pairing
    g = torch.Generator().manual_seed(5)
    hold = torch.randint(0, pool.n_items, (B,), generator=g)
    himg = pool.images[hold.cuda()].contiguous()
    clip = torch.randint(0, pool.clips.shape[0], (2 * B,), generator=g)
    hl = torch.full((2 * B,), 16000, dtype=torch.int32)
    hfe = cpu_features(pool, clip, hl)
    m.eval()
    with torch.no_grad():
        d = m(himg, hfe[:B].cuda(), hfe[B:].cuda())
        ref.model.eval()
        a, p, n = ref.model((himg.cpu() / 255.).float(), hfe[:B], hfe[B:])
    for k, want in (("image_feat", a), ("sound_feat_positive", p), ("sound_feat_negative", n)):
        err = float((d[k].cpu() - want).abs().max())
        assert err < 1e-3, (k, err)
    # (the PARAMETERS after the 10 steps are judged against a float64 run of the same steps, with torch's own fp32 run as
    #  the yardstick: tests/test_gpu_round3.py::test_config2_parameter_trajectory_against_a_float64_yardstick -- the
    #  16-thread-vs-1-thread self-drift bound that stood here in round 2 was 4.6x wider than what it measured)


def test_ragged_epoch_300_at_batch_128(var_amd, golden_dir):
    """The reference's default: 300 triplets, batch 128, drop_last=False -> 128 / 128 / 44 (VAR/pretext_VAR.py:24,
    fourInARow/config.py:25,37).  Two epochs through the replayed step (second graph for the short batch) equal eager
    step_from_dataset calls on the same rows, the short batch averaged over 44."""
    sd = load(golden_dir, "kuka_weights.npz")
    B = 128
    pool = var_amd.SyntheticTripletPool(300, hw=84, seed=31, clips_per_class=4).freeze_pairs()
    spe, bt = pool.steps_per_epoch(B), pool.tail_batch(B)
    assert (spe, bt) == (3, 44)
    table = pool.index_table(B, 2 * spe, drop_last=False)
    assert table.shape == (6, 5 * B)
    mb = make_model(var_amd, sd)
    tb = var_amd.VARTrainer(mb, lr=1e-3)
    replay, _ = tb.capture_epoch_steps(pool.images, pool.clips, B, table, steps_per_epoch=spe, tail_batch=bt)
    losses_b = [float(replay().item()) for _ in range(2 * spe)]
    pb = mb.flat_parameters().cpu().numpy().copy()
    ma = make_model(var_amd, sd)
    ta = var_amd.VARTrainer(ma, lr=1e-3)
    losses_a = []
    for s in range(2 * spe):
        Bs = bt if s % spe == spe - 1 else B
        r = table[s]
        losses_a.append(float(ta.step_from_dataset(pool.images, r[:Bs].contiguous(), pool.clips, r[Bs:3 * Bs].contiguous(),
                                                   r[3 * Bs:5 * Bs].contiguous()).item()))
    assert np.allclose(losses_a, losses_b, rtol=0, atol=1e-6), (losses_a, losses_b)
    pa = ma.flat_parameters().cpu().numpy()
    assert np.mean(np.abs(pa - pb) < 2e-6) > 0.995
    # and the short batch against the oracle: mean over 44, not over 128
    r = table[2].cpu()
    mo = make_model(var_amd, sd)
    to = var_amd.VARTrainer(mo)
    to.step_from_dataset(pool.images, table[2, :bt].contiguous(), pool.clips, table[2, bt:3 * bt].contiguous(),
                         table[2, 3 * bt:5 * bt].contiguous())
    l_ref, _, _ = orc.loss_grad(orc.flatten_params(sd), pool.images[r[:bt].long()].cpu().numpy(),
                                cpu_features(pool, r[bt:2 * bt], r[3 * bt:4 * bt]).numpy(),
                                cpu_features(pool, r[2 * bt:3 * bt], r[4 * bt:5 * bt]).numpy())
    assert abs(to.loss.item() - l_ref) < 2e-5


def test_two_models_interleaved_on_one_device(var_amd, golden_dir):
    """A training model and a frozen copy with DIFFERENT weights on one GPU (the reference's RL stage,
    Envs/vec_env/vec_pretext_normalize.py:82-94): each keeps its own packed weight image, so interleaved calls stay
    correct, and a forward re-packs only when the parameters changed."""
    sd1, sd2 = load(golden_dir, "kuka_weights.npz"), load(golden_dir, "kuka_weights2.npz")
    fx1, fx2 = load(golden_dir, "kuka_h84.npz"), load(golden_dir, "kuka_h84_w2.npz")
    m1, m2 = make_model(var_amd, sd1), make_model(var_amd, sd2)
    tr = var_amd.VARTrainer(m1, lr=0.0)
    args1 = (cuda(fx1['image']), cuda(fx1['sound_positive']), cuda(fx1['sound_negative']))
    args2 = (cuda(fx2['image']), cuda(fx2['sound_positive']), cuda(fx2['sound_negative']))
    for _ in range(2):
        tr.loss_and_grads(*args1)                                  # the trainer never re-packs by itself
        assert abs(tr.loss.item() - float(fx1['loss'])) < 1e-5
        with torch.no_grad():
            d2 = m2(*args2)
        assert np.max(np.abs(d2['image_feat'].cpu().numpy() - fx2['image_feat'])) < 1e-5
    g1 = orc.unflatten_params(tr.grads.cpu().numpy())
    for k, _ in orc.PARAM_SPECS:
        ref = fx1['grad.' + k]
        assert np.max(np.abs(g1[k] - ref)) / (np.max(np.abs(ref)) + 1e-30) < 1e-3, k
    # no re-pack on an unchanged model; a re-pack after the parameters change
    w2 = m2.hip_weights()
    key = w2.key
    with torch.no_grad():
        m2(*args2)
    assert m2.hip_weights().key == key
    with torch.no_grad():
        m2.imgTriplet[2].bias.add_(1.0)
        d3 = m2(*args2)
    assert m2.hip_weights().key != key
    assert not torch.equal(d3['image_feat'], d2['image_feat'])
    # a stale binding is refused by the library itself
    c = tr.ctx
    from var_amd._lib import ptr
    m2.hip_weights()                                               # m2's image bound ...
    rc = c.lib.var_arm_loss_grad(c.handle, c.stream(), ptr(m1.flat_parameters()), ptr(args1[0]), 1, args1[0].stride(0),
                                 ptr(args1[1]), ptr(args1[2]), 4, 84, 1.0, 0.25, ptr(tr.gbuf), tr._loss_ptr(), None)
    assert rc != 0 and b"weight image" in c.lib.var_last_error(c.handle)   # ... m1's parameters given


def test_backward_of_an_overwritten_forward_raises(var_amd, golden_dir):
    sd = load(golden_dir, "kuka_weights.npz")
    fx = load(golden_dir, "kuka_h84.npz")
    m1, m2 = make_model(var_amd, sd), make_model(var_amd, sd)
    args = (cuda(fx['image']), cuda(fx['sound_positive']), cuda(fx['sound_negative']))
    d1 = m1(*args)
    m2(*args)                                                      # overwrites the one saved forward of the context
    loss = torch.nn.TripletMarginLoss()(d1['image_feat'], d1['sound_feat_positive'], d1['sound_feat_negative'])
    with pytest.raises(var_amd.VarHipError, match="saved forward"):
        loss.backward()


@pytest.mark.parametrize("mode", ["accumulate", "zero_grad_keep"])
def test_autograd_gradient_accumulation(var_amd, golden_dir, mode):
    """Two backwards with different batches: without zero_grad the .grad tensors hold the SUM; with
    zero_grad(set_to_none=False) (what torch 1.x's model.zero_grad() did, VAR/pretext_VAR.py:56) the second gradient
    alone -- the backward returns fresh tensors, never views of a buffer it will overwrite."""
    sd = load(golden_dir, "kuka_weights.npz")
    fxa, fxb = load(golden_dir, "kuka_h84.npz"), load(golden_dir, "kuka_edge.npz")
    m = make_model(var_amd, sd)
    m.train()
    crit = torch.nn.TripletMarginLoss(margin=1.0, p=2)
    batches = [(fxa['image'], fxa['sound_positive'], fxa['sound_negative']),
               (fxb['image'], fxb['sound_positive'], fxb['sound_negative'])]
    refs = [orc.loss_grad(orc.flatten_params(sd), *b)[1] for b in batches]
    for i, b in enumerate(batches):
        if i == 1 and mode == "zero_grad_keep":
            m.zero_grad(set_to_none=False)
        d = m(cuda(b[0]), cuda(b[1]), cuda(b[2]))
        crit(d['image_feat'], d['sound_feat_positive'], d['sound_feat_negative']).backward()
    want = refs[1] if mode == "zero_grad_keep" else refs[0] + refs[1]
    got = torch.cat([dict(m.named_parameters())[k].grad.reshape(-1) for k, _ in var_amd.PARAM_SPECS]).cpu().numpy()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-4


def test_captured_graphs_survive_a_replan(var_amd, golden_dir):
    """IntrinsicReward.capture(8) bakes workspace addresses into its graphs; a later, larger plan (a training step at
    batch 32, or another image size) must not invalidate them: superseded workspaces stay allocated."""
    sd = load(golden_dir, "kuka_weights.npz")
    fx = load(golden_dir, "kuka_edge.npz")
    frozen = make_model(var_amd, sd)
    # a fresh context state is not guaranteed (other tests planned larger batches): force a small plan first
    ir = var_amd.IntrinsicReward(frozen).capture(4)
    img = cuda(fx['image'])
    a0 = [t.clone() for t in ir.step(img, cuda(fx['sound_positive']))]
    from var_amd._lib import Context
    ctx = Context.get(0)
    gen0 = ctx.lib.var_plan_generation(ctx.handle)
    # grows past every earlier plan of the process (a 96 x 96 plan holds more per image than an 84 x 84 one: twice the batch is enough): re-plans
    grown = 2 * ctx.plan[0] + 64
    ctx.check(ctx.lib.var_plan(ctx.handle, grown, 84), "var_plan")
    ctx.plan = (grown, 84)
    assert ctx.lib.var_plan_generation(ctx.handle) == gen0 + 1
    trainer_model = make_model(var_amd, load(golden_dir, "kuka_weights2.npz"))
    tr = var_amd.VARTrainer(trainer_model)
    B = 32
    rng = np.random.default_rng(0)
    tr.step(cuda(rng.integers(0, 256, (B, 3, 84, 84), dtype=np.uint8)), cuda(rng.standard_normal((B, 1, 100, 40)).astype(np.float32)),
            cuda(rng.standard_normal((B, 1, 100, 40)).astype(np.float32)))
    a1 = ir.step(img, cuda(fx['sound_positive']))
    torch.cuda.synchronize()
    for x, y in zip(a0, a1):
        assert torch.equal(x, y)
    assert np.max(np.abs(a1[0].cpu().numpy() - fx['a.image_feat'])) < 1e-4


def test_inbatch_head_at_config3_size(var_amd):
    """BASELINE configs[2]: 256 local anchors against the 4096 candidate sound embeddings of a 2048-triplet global batch
    (8 ranks x [256 positives ; 256 negatives]): loss and both gradients against the torch restatement."""
    from oracle.torch_oracle import inbatch_contrastive_loss as ref_loss
    rng = np.random.default_rng(8)
    B, M, rank = 256, 4096, 3
    unit = lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)  # noqa: E731
    a, cand = unit(rng.standard_normal((B, 3))), unit(rng.standard_normal((M, 3)))
    target = (np.arange(B) + rank * 2 * B).astype(np.int32)
    loss, ga, gc = var_amd.inbatch_contrastive_loss(cuda(a), cuda(cand), cuda(target), tau=0.1, inv_count=1.0 / 2048)
    ta, tc = torch.from_numpy(a).double().requires_grad_(), torch.from_numpy(cand).double().requires_grad_()
    rl = ref_loss(ta, tc, torch.from_numpy(target).long(), tau=0.1, inv_count=1.0 / 2048)
    rl.backward()
    assert abs(loss.item() - rl.item()) < 1e-5 * max(1.0, abs(rl.item()))
    np.testing.assert_allclose(ga.cpu().numpy(), ta.grad.numpy(), atol=1e-7, rtol=2e-4)
    np.testing.assert_allclose(gc.cpu().numpy(), tc.grad.numpy(), atol=1e-7, rtol=2e-4)


def test_replayed_inbatch_step_equals_eager(var_amd, golden_dir):
    """The replayed in-batch-negatives step (bench.py --head inbatch: gather, MFCC, encoder, head, backward, Adam and the
    row fetch in ONE graph) against eager step_inbatch calls on the same rows."""
    sd = load(golden_dir, "kuka_weights.npz")
    B = 32
    pool = var_amd.SyntheticTripletPool(128, hw=84, seed=41, clips_per_class=4).freeze_pairs()
    table = pool.index_table(B, 3, drop_last=True)[:3].contiguous()
    mb = make_model(var_amd, sd)
    tb = var_amd.VARTrainer(mb, lr=1e-3)
    replay, _ = tb.capture_inbatch_epoch_steps(pool.images, pool.clips, B, table, tau=0.1)
    losses_b = [float(replay().item()) for _ in range(4)]
    ma = make_model(var_amd, sd)
    ta = var_amd.VARTrainer(ma, lr=1e-3)
    losses_a = []
    for s in range(4):
        r = table[s % 3]
        f = var_amd.mfcc(pool.clips, r[3 * B:], out_frames=100, clip_index=r[B:3 * B])
        losses_a.append(float(ta.step_inbatch(pool.images[r[:B].long()].contiguous(), f[:B].contiguous(), f[B:].contiguous(),
                                              tau=0.1).item()))
    assert np.allclose(losses_a, losses_b, rtol=0, atol=1e-5), (losses_a, losses_b)
    assert float((ma.flat_parameters() - mb.flat_parameters()).abs().max()) < 1e-4


@pytest.mark.parametrize("dataset", ["NSynth", "UrbanSound", "GoogleCommand"])
def test_mfcc_with_the_dataset_stft_parameters(var_amd, dataset):
    """Envs/audioLoader.py:23-31: NSynth / UrbanSound clips use n_fft 1024, a 50 ms window and a 40 ms step (26 frames per
    second of audio); GoogleCommand 512 / 25 ms / 10 ms.  var_mfcc_ex against the numpy oracle with the same parameters,
    incl. a short clip, an odd length and the "empty" class; odd row strides take the per-sample path."""
    n_fft, win, hop = mfcc_np.DATASET_STFT[dataset]
    clips = mfcc_np.synth_clips(5, seed=33)
    lens = np.array([16000, 9001, 700, 0, 16000], dtype=np.int32)
    frames = 1 + 16000 // hop
    out = var_amd.mfcc(cuda(clips), cuda(lens), frames + 3, dataset=dataset).cpu().numpy()
    assert out.shape == (5, 1, frames + 3, 40)
    for i in range(5):
        if lens[i] == 0:
            assert np.all(out[i] == 0)
            continue
        ref = mfcc_np.mfcc_torchaudio(clips[i, :lens[i]], n_fft=n_fft, win=win, hop=hop)
        T = ref.shape[0]
        assert T == 1 + lens[i] // hop
        assert np.max(np.abs(out[i, 0, :T] - ref)) < 2e-3, (i, np.max(np.abs(out[i, 0, :T] - ref)))
        assert np.all(out[i, 0, T:] == 0)                          # MFCC-domain zero padding (audioLoader.py:245-250)
    # odd row stride: rows are only 2-byte aligned
    odd = np.zeros((3, 16001), np.int16)
    odd[:, :16000] = clips[:3]
    got = var_amd.mfcc(cuda(odd), cuda(np.array([16000, 16000, 12345], np.int32)), 100, dataset=dataset).cpu().numpy()
    for i, n in enumerate((16000, 16000, 12345)):
        ref = mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(odd[i, :n], n_fft=n_fft, win=win, hop=hop))
        assert np.max(np.abs(got[i] - ref)) < 2e-3
    with pytest.raises(var_amd.VarHipError):
        var_amd.mfcc(cuda(clips), cuda(lens), 100, n_fft=1000)


def test_pool_training_loop_on_the_gpu(var_amd, tmp_path):
    """train_representation_from_pool at the reference's default shape -- 300 triplets, batch 128 (128 / 128 / 44), per-epoch
    MultiStepLR -- learns (the triplet loss of a memorisable pool falls), and its legacy .pt loads into a fresh model that
    reproduces the trained embeddings."""
    torch.manual_seed(453)
    model = var_amd.VARPretextNet(cfg()).to("cuda")
    pool = var_amd.SyntheticTripletPool(300, hw=84, seed=7, clips_per_class=2, empty_frac=0.1).freeze_pairs()
    out = var_amd.train_representation_from_pool(model, pool, 12, 128, lr=1e-3, milestones=(8,), gamma=0.2,
                                                 save_dir=str(tmp_path), save_interval=5, log=lambda *a: None)
    assert len(out) == 12 and np.all(np.isfinite(out))
    assert out[-1] < out[0] - 0.1, out
    assert sorted(p.name for p in tmp_path.iterdir()) == ['11.pt', '4.pt', '9.pt', 'progress.csv']
    fresh = var_amd.VARPretextNet(cfg()).to("cuda")
    fresh.load_state_dict(torch.load(tmp_path / '11.pt', weights_only=True))
    row = pool.epoch_index_table(64)[0]
    img = pool.images[row[:64].long()].contiguous()
    f = var_amd.mfcc(pool.clips, row[3 * 64:5 * 64], 100, row[64:3 * 64])
    with torch.no_grad():
        a = model(img, f[:64].contiguous(), None)
        b = fresh(img, f[:64].contiguous(), None)
    assert torch.equal(a['image_feat'], b['image_feat']) and torch.equal(a['sound_feat_positive'], b['sound_feat_positive'])


def test_trainer_notices_parameters_changed_behind_its_back(var_amd, golden_dir):
    """load_state_dict between steps (what a fine-tune run does) without an explicit tr.pack(): the next eager step and
    the next replay re-pack the weight image by themselves (torch's version counters) and use the NEW weights."""
    sd1, sd2 = load(golden_dir, "kuka_weights.npz"), load(golden_dir, "kuka_weights2.npz")
    fx2 = load(golden_dir, "kuka_h84_w2.npz")
    m = make_model(var_amd, sd1)
    tr = var_amd.VARTrainer(m, lr=0.0)
    args = (cuda(fx2['image']), cuda(fx2['sound_positive']), cuda(fx2['sound_negative']))
    tr.loss_and_grads(*args)
    assert abs(tr.loss.item() - float(fx2['loss'])) > 1e-4          # weights 1 on fixture 2: some other loss
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    tr.loss_and_grads(*args)
    assert abs(tr.loss.item() - float(fx2['loss'])) < 1e-5          # weights 2 without tr.pack()
    # the same through a replayed step
    pool = var_amd.SyntheticTripletPool(32, hw=84, seed=2, clips_per_class=2).freeze_pairs()
    table = pool.index_table(16, 2, drop_last=True)[:2].contiguous()
    replay, load_table = tr.capture_epoch_steps(pool.images, pool.clips, 16, table)
    l2 = float(replay().item())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd1.items()})
    load_table(table)
    l1 = float(replay().item())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    load_table(table)
    assert abs(float(replay().item()) - l2) < 1e-6 and abs(l1 - l2) > 1e-5
