"""GPU tests added in round 3 (run with -m gpu on an MI355X): what the round-2 review asked for -- the bf16 mode at the size
it is benchmarked at, a parameter-trajectory bound against a float64 yardstick, the optimiser's behaviour after a timed-out
persistent GRU launch, edits of the parameters behind torch's version counters."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mfcc_np  # noqa: E402  (checker only)
from oracle.torch_oracle import CPUTrainer, IthorNetCPU, KukaNetCPU  # noqa: E402


def cfg(h=84):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 100, 40), representationDim=3)


def icfg(h=96):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 600, 40), representationDim=3)


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def make_model(var_amd, sd, h=84):
    m = var_amd.VARPretextNet(cfg(h))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to("cuda")


def cpu_features(pool, clip_idx, lens):
    out = np.zeros((len(clip_idx), 1, 100, 40), np.float32)
    clips = pool.clips.cpu().numpy()
    for i, (c, n) in enumerate(zip(clip_idx.tolist(), lens.tolist())):
        if n > 0:
            out[i] = mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(clips[c, :n]).astype(np.float32))
    return torch.from_numpy(out)


# ------------------------------------------------------------------------------------------------------------------
# parameters edited through .data (no version counter moves): the eager forward must still see them
# ------------------------------------------------------------------------------------------------------------------
def test_forward_sees_parameter_edits_made_through_dot_data(var_amd, golden_dir):
    sd = load(golden_dir, "kuka_weights.npz")
    fx = load(golden_dir, "kuka_h84.npz")
    m = make_model(var_amd, sd)
    args = tuple(torch.from_numpy(fx[k]).cuda() for k in ("image", "sound_positive", "sound_negative"))
    with torch.no_grad():
        d0 = m(*args)
    before = (m.flat_parameters()._version, sum(p._version for p in m.parameters()))
    for p in m.parameters():
        p.data.mul_(0.5)                                    # bumps neither counter (the round-2 advisor's case)
    torch.nn.init.constant_(m.imgTriplet[2].bias.data, 0.25)
    assert (m.flat_parameters()._version, sum(p._version for p in m.parameters())) == before
    with torch.no_grad():
        d1 = m(*args)
    ref = KukaNetCPU(84)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        a, p_, n_ = ref((torch.from_numpy(fx["image"]) / 255.).float(), torch.from_numpy(fx["sound_positive"]),
                        torch.from_numpy(fx["sound_negative"]))
    assert not torch.equal(d1["image_feat"], d0["image_feat"])
    for k, want in (("image_feat", a), ("sound_feat_positive", p_), ("sound_feat_negative", n_)):
        assert float((d1[k].cpu() - want).abs().max()) < 1e-5, k
    # the explicit form for code that binds the packed image itself
    w = m.pack()
    assert w is m.hip_weights()


# ------------------------------------------------------------------------------------------------------------------
# trajectory: HIP vs float64, with torch-fp32 vs float64 as the yardstick
# ------------------------------------------------------------------------------------------------------------------
def test_config2_parameter_trajectory_against_a_float64_yardstick(var_amd, golden_dir):
    """10 optimisation steps at batch 256 (BASELINE configs[1]) three times on the same fp32 inputs and initial weights:
    the HIP replayed graph, torch fp32 on the CPU, torch float64 on the CPU.  Rounding moves Adam's normalised update by
    percents near this initialisation, so the yardstick is not a constant: the HIP parameters must be no further from
    the float64 run than 2x what torch's own fp32 run is -- in the mean, at the 99th percentile and at the 99.9th."""
    sd = load(golden_dir, "kuka_weights.npz")
    B, steps = 256, 10
    pool = var_amd.SyntheticTripletPool(768, hw=84, seed=21, clips_per_class=3).freeze_pairs()
    table = pool.index_table(B, steps, drop_last=True)[:steps].contiguous()
    m = make_model(var_amd, sd)
    tr = var_amd.VARTrainer(m, lr=1e-4, weight_decay=1e-6)
    replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
    r32 = CPUTrainer(state_dict=sd, lr=1e-4, weight_decay=1e-6)
    r64 = CPUTrainer(state_dict=sd, lr=1e-4, weight_decay=1e-6, dtype=torch.float64)
    cache = {}

    def feats_of(idx, lens):
        key = (tuple(idx.tolist()), tuple(lens.tolist()))
        if key not in cache:
            cache[key] = cpu_features(pool, idx, lens)
        return cache[key]
    tcpu = table.cpu()
    for s in range(steps):
        r = tcpu[s]
        img = pool.images[r[:B].long()].cpu()
        fp, fn = feats_of(r[B:2 * B], r[3 * B:4 * B]), feats_of(r[2 * B:3 * B], r[4 * B:])
        l64 = r64.step(img, fp, fn)
        l32 = r32.step(img, fp, fn)
        got = float(replay().item())
        assert abs(got - l64) < 2e-5 * (s + 1), (s, got, l64)
        assert abs(got - l64) <= 2.0 * abs(l32 - l64) + 2e-6 * (s + 1), (s, got, l32, l64)
    flat_of = lambda t: torch.cat([t.model.state_dict()[k].reshape(-1).double() for k, _ in var_amd.PARAM_SPECS])  # noqa: E731
    truth = flat_of(r64)
    d_hip = (m.flat_parameters().cpu().double() - truth).abs()
    d_t32 = (flat_of(r32) - truth).abs()
    q = lambda t, f: float(torch.quantile(t, f))           # noqa: E731
    print("after 10 steps, |param - float64 run|: HIP mean %.3e p99 %.3e p99.9 %.3e max %.3e | torch fp32 mean %.3e p99 %.3e "
          "p99.9 %.3e max %.3e" % (d_hip.mean(), q(d_hip, .99), q(d_hip, .999), d_hip.max(),
                                   d_t32.mean(), q(d_t32, .99), q(d_t32, .999), d_t32.max()))
    assert float(d_hip.mean()) <= 2.0 * float(d_t32.mean()) + 1e-9
    assert q(d_hip, .99) <= 2.0 * q(d_t32, .99) + 1e-8
    assert q(d_hip, .999) <= 2.0 * q(d_t32, .999) + 1e-7
    # and an absolute frame for the numbers above: both fp32 runs stay within a few percent of the distance Adam can
    # travel in 10 steps (1e-3) at the 99.9th percentile
    assert q(d_hip, .999) < 0.1 * steps * 1e-4


# ------------------------------------------------------------------------------------------------------------------
# iTHOR bf16 mode at the size the bench runs it at: B = 256 -> 512 clips -> 256 persistent GRU workgroups
# ------------------------------------------------------------------------------------------------------------------
def _ithor_batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, generator=g)
    snd = torch.randn(2 * B, 1, 600, 40, generator=g) * 3
    snd[1, :, 350:] = 0.0                                   # a zero-padded clip, as processSoundFeat leaves short ones
    return img, snd[:B].contiguous(), snd[B:].contiguous()


def _param_spans(m):
    spans, o = {}, 0
    for k, prm in m.named_parameters():
        spans[k] = (o, o + prm.numel())
        o += prm.numel()
    return spans


def test_ithor_bf16_at_the_benchmarked_batch_256(var_amd):
    """(1) the persistent GRU launches (256 workgroups on 256 CUs: the full-residency edge) against the per-step launches:
    identical loss and gradients (GRU bias gradients: summation order only), no hand-off timed out, deterministic;
    (2) bf16 vs fp32 on the same batch: loss, embeddings, gradient within the drift bounds of the small-batch tests;
    (3) the fp32 path against the CPU oracle on a 16-sample slice of the same batch."""
    import ctypes
    from var_amd._lib import Context
    B = 256
    img, pos, neg = (t.cuda() for t in _ithor_batch(B, 41))
    torch.manual_seed(977)
    ref = IthorNetCPU()
    sd = ref.state_dict()
    out = {}
    for prec in ("fp32", "bf16"):
        m = var_amd.IthorVARPretextNet(icfg(96))
        m.load_state_dict(sd)
        m = m.to("cuda").set_precision(prec)
        tr = var_amd.IthorTrainer(m)
        loss, feats = tr.loss_and_grads(img, pos, neg, feats=True)
        torch.cuda.synchronize()
        out[prec] = (float(loss), feats.clone(), tr.grads.clone())
        if prec == "bf16":
            ctx = Context.get(0)
            assert m.gru_status() == 0
            per = {}
            try:
                for mode in (0, 1, 1):
                    assert ctx.lib.var_ithor_set_gru_sequence(ctx.handle, mode) >= 0
                    l, _ = tr.loss_and_grads(img, pos, neg)
                    torch.cuda.synchronize()
                    per.setdefault(mode, []).append((float(l), tr.grads.clone()))
            finally:
                ctx.lib.var_ithor_set_gru_sequence(ctx.handle, 1)
            word = ctypes.c_uint(123)
            assert ctx.lib.var_ithor_gru_status(ctx.handle, ctypes.byref(word)) == 0 and word.value == 0
            (l0, g0), = per[0]
            spans = _param_spans(m)
            bias = [k for k in spans if k.startswith("rnn.bias")]
            for l1, g1 in per[1]:
                assert torch.isfinite(g1).all() and l1 == l0 == out["bf16"][0]
                a, b = g1.clone(), g0.clone()
                for k in bias:
                    lo, hi = spans[k]
                    assert float((a[lo:hi] - b[lo:hi]).abs().max()) <= 2e-6 * float(b[lo:hi].abs().max()) + 1e-12, k
                    a[lo:hi] = 0
                    b[lo:hi] = 0
                assert torch.equal(a, b), float((a - b).abs().max())
            assert torch.equal(per[1][0][1], per[1][1][1])
            assert torch.equal(per[1][0][1], out["bf16"][2])
    l32, f32, g32 = out["fp32"]
    l16, f16, g16 = out["bf16"]
    assert abs(l16 - l32) < 2e-3, (l16, l32)
    assert float((f16 - f32).abs().max()) < 1e-2
    l2 = lambda a, b: float((a - b).norm() / b.norm())       # noqa: E731
    assert l2(g16, g32) < 0.15, l2(g16, g32)
    # (3) 16 samples of the batch through the CPU oracle: a sample's embeddings do not depend on the rest of the batch
    sl = slice(64, 80)
    with torch.no_grad():
        a, p_, n_ = ref((img[sl].cpu() / 255.).float(), pos[sl].cpu(), neg[sl].cpu())
    want = torch.cat([a, p_, n_], 1)
    assert float((f32[sl].cpu() - want).abs().max()) < 1e-4
    assert float((f16[sl].cpu() - want).abs().max()) < 1e-2


def test_ithor_bf16_gradient_drift_is_accounted_for(var_amd):
    """Where the bf16 gradient leaves the fp32 one, the cause must be visible in the workspace, tensor by tensor.
    (a) Routing: a ReLU gate (the two wide sound convolutions, the image branch's five convolutions) that differs between the
        two forwards must sit at a pre-activation within bf16 rounding of zero, and such units are a small fraction.
    (b) Weight tensors: within 30 % in L2, and any drift above 3 % needs at least one differing gate.
    (c) The image branch's BIAS gradients (32-128 sums over every pixel of the batch) drifted by up to 34 % in round 3 and the
        bound had been widened to 50 % on the strength of a comment about pool winners.  Traced now (tests/diagnostics/bf16_bias_trace.py):
        the terms of those sums are the buffers ga1..ga5; split by unit, the part of the difference on units the gradient
        reaches in only ONE of the two runs (gate / pool-winner flips) is 2-15 % of the tensor's norm -- the rest sits on
        units BOTH runs reach and is operand rounding: the sums cancel heavily (sum |g| / |sum g| = 8 ... 270), so a
        per-term relative perturbation of 2^-10 ... 2^-8 (bf16 operands upstream) shows up multiplied by that factor.  The
        bound is therefore drift <= 2^-7 x (sum |g| / |sum g|), computed from the fp32 run's own terms, and the routed part
        alone must stay under 20 %."""
    from var_amd._lib import Context
    B = 8
    img, pos, neg = (t.cuda() for t in _ithor_batch(B, 43))
    torch.manual_seed(977)
    sd = IthorNetCPU().state_dict()
    side = {1: 96, 2: 96, 3: 48, 4: 24, 5: 12}                  # output map of image conv l
    ich = [3, 32, 32, 64, 64, 128]
    acts, grads, terms = {}, {}, {}
    for prec in ("fp32", "bf16"):
        m = var_amd.IthorVARPretextNet(icfg(96))
        m.load_state_dict(sd)
        m = m.to("cuda").set_precision(prec, keep_fp32_activations=True)
        tr = var_amd.IthorTrainer(m)
        tr.loss_and_grads(img, pos, neg)
        torch.cuda.synchronize()
        ctx = Context.get(0)
        n = 2 * B
        acts[prec] = {"s1": ctx.debug_buffer("ithor_s1")[:n * 64 * 300 * 20].clone(),
                      "s2": ctx.debug_buffer("ithor_s2")[:n * 64 * 150 * 13].clone()}
        terms[prec] = {}
        for l in range(1, 6):
            cnt = B * ich[l] * side[l] ** 2
            acts[prec][f"a{l}"] = ctx.debug_buffer(f"ithor_a{l}")[:cnt].clone()
            terms[prec][l] = ctx.debug_buffer(f"ithor_ga{l}")[:cnt].clone().view(B, ich[l], -1)      # d loss / d pre-activation
        grads[prec] = tr.grads.clone()
        spans = _param_spans(m)
    nflip = 0
    for k in acts["fp32"]:
        a32, a16 = acts["fp32"][k], acts["bf16"][k]
        scale = float(a32.abs().max())
        flip = (a32 > 0) != (a16 > 0)
        nflip += int(flip.sum())
        frac = float(flip.float().mean())
        assert frac < 0.01, (k, frac)
        if flip.any():
            survivor = torch.maximum(a32[flip], a16[flip])
            assert float(survivor.max()) < 2e-2 * scale, (k, float(survivor.max()), scale)
        # away from the gates the two activations agree to bf16 operand rounding
        assert float((a32 - a16).abs().max()) < 3e-2 * scale, k
    img_bias = {f"imgBranch.{i}.bias": l for l, i in zip(range(1, 6), (0, 2, 5, 8, 11))}
    worst = {}
    for k, (lo, hi) in spans.items():
        g32, g16 = grads["fp32"][lo:hi], grads["bf16"][lo:hi]
        d = float((g16 - g32).norm() / (g32.norm() + 1e-30))
        worst[k] = d
        if k in img_bias:
            t32, t16 = terms["fp32"][img_bias[k]], terms["bf16"][img_bias[k]]
            assert float((t32.sum((0, 2)) - g32).norm()) <= 1e-5 * float(g32.norm()), k       # the buffers ARE the sums' terms
            kappa = float(t32.abs().sum((0, 2)).norm() / g32.norm())
            one_sided = (t32 != 0) != (t16 != 0)                  # units the gradient reaches in one run only
            diff = t16 - t32
            routed = float(torch.where(one_sided, diff, torch.zeros_like(diff)).sum((0, 2)).norm() / g32.norm())
            assert float(one_sided.float().mean()) < 0.01 and routed < 0.2, (k, float(one_sided.float().mean()), routed)
            assert d < 2.0 ** -7 * kappa, (k, d, kappa)
            print(f"{k}: drift {d:.3f} = routing {routed:.3f} + rounding x cancellation {kappa:.0f}")
        else:
            assert d < (0.3 if hi - lo >= 1024 else 0.5), (k, d)
    drifted = {k: v for k, v in worst.items() if v > 0.03 and k not in img_bias}
    if drifted:
        assert nflip > 0, f"gradient tensors drift by more than 3 % in L2 without a single differing gate: {drifted}"
    print("gates that differ:", nflip, "| worst per-tensor L2 drift:", max(worst.values()))


# ------------------------------------------------------------------------------------------------------------------
# a timed-out persistent GRU launch: the optimiser must skip that step, the next one must run
# ------------------------------------------------------------------------------------------------------------------
def test_gru_time_out_skips_the_optimiser_step_and_clears_itself(var_amd):
    from var_amd._lib import Context
    B = 3
    img, pos, neg = (t.cuda() for t in _ithor_batch(B, 45))
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(icfg(96)).to("cuda").set_precision("bf16")
    tr = var_amd.IthorTrainer(m)
    tr.step(img, pos, neg)                                  # a good step (plans the context, moves the moments off zero)
    torch.cuda.synchronize()
    assert m.gru_status() == 0
    p0, m0, v0 = m.flat_parameters().clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone()
    ctx = Context.get(0)
    assert ctx.lib.var_debug_ithor_gru_drop_workgroup(ctx.handle) == 0
    loss = tr.step(img, pos, neg)                           # the forward's persistent launch misses a workgroup: time-out
    torch.cuda.synchronize()
    assert not np.isfinite(float(loss)) and m.gru_status() != 0
    assert torch.equal(m.flat_parameters(), p0) and torch.equal(tr.exp_avg, m0) and torch.equal(tr.exp_avg_sq, v0)
    loss = tr.step(img, pos, neg)                           # the next step clears the word by itself and trains
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    assert bool(torch.isfinite(m.flat_parameters()).all()) and not torch.equal(m.flat_parameters(), p0)
    assert m.gru_status() & 0x40000000                      # the earlier time-out stays on record (sticky word)
    # the training loop reports it once and takes the per-step launches from there on
    logs = []
    assert ctx.lib.var_debug_ithor_gru_drop_workgroup(ctx.handle) == 0
    losses = var_amd.train_representation(m, lambda: iter([(img, pos, neg, None)] * 3), epochs=2, log=lambda *a: logs.append(a))
    assert all(np.isfinite(v) for v in losses)
    assert any("timed out" in str(a[0]) for a in logs)
    assert bool(torch.isfinite(m.flat_parameters()).all())


def test_one_replica_timing_out_makes_every_replica_skip_the_step(var_amd):
    """Data parallelism (round-3 advisor finding): the time-out word is per rank, but the NaN gradient of the rank that timed
    out is summed into EVERY rank's buffer.  Two replicas on one device, the all-reduce done by hand (sum of the two
    [gradient | loss] buffers, exactly what IthorTrainer.allreduce does): replica A's persistent GRU launch times out, B's
    runs.  Both must skip the step -- parameters, moments AND the applied-step count untouched, replicas identical -- through
    the loss guard (var_ithor_guard_loss: B's own time-out word is clean, and A's was cleared by B's forward on the shared
    workspace before A's Adam runs).  The step after that trains both, identically."""
    from var_amd._lib import Context
    B = 3
    batches = [tuple(t.cuda() for t in _ithor_batch(B, 50 + r)) for r in range(2)]
    reps = []
    for r in range(2):
        torch.manual_seed(5)
        m = var_amd.IthorVARPretextNet(icfg(96)).to("cuda").set_precision("bf16")
        reps.append((m, var_amd.IthorTrainer(m)))
    ctx = Context.get(0)

    def dp_step(drop_rank=None):
        for r, (m, tr) in enumerate(reps):
            if r == drop_rank:
                assert ctx.lib.var_debug_ithor_gru_drop_workgroup(ctx.handle) == 0
            tr.loss_and_grads(*batches[r], global_batch=2 * B)
        total = reps[0][1].gbuf + reps[1][1].gbuf                # the all-reduce (SUM) of [gradients | loss]
        for _, tr in reps:
            tr.gbuf.copy_(total)
            tr.adam()
        torch.cuda.synchronize()
        return float(total[-1])

    assert np.isfinite(dp_step())                                # a good step first (moments off zero)
    assert torch.equal(reps[0][0].flat_parameters(), reps[1][0].flat_parameters())
    snap = [(m.flat_parameters().clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone()) for m, tr in reps]
    assert [tr.step_count for _, tr in reps] == [1, 1]
    assert not np.isfinite(dp_step(drop_rank=0))                 # replica A times out: the summed loss is NaN on both
    for (m, tr), (p0, m0, v0) in zip(reps, snap):
        assert torch.equal(m.flat_parameters(), p0) and torch.equal(tr.exp_avg, m0) and torch.equal(tr.exp_avg_sq, v0)
        assert tr.step_count == 1                                # the bias corrections do not drift
    assert np.isfinite(dp_step())
    assert torch.equal(reps[0][0].flat_parameters(), reps[1][0].flat_parameters())
    assert bool(torch.isfinite(reps[0][0].flat_parameters()).all()) and not torch.equal(reps[0][0].flat_parameters(), snap[0][0])
    assert [tr.step_count for _, tr in reps] == [2, 2]


# ------------------------------------------------------------------------------------------------------------------
# iTHOR model over an HBM-resident pool: replayed ragged epochs = eager steps on the same rows
# ------------------------------------------------------------------------------------------------------------------
def test_ithor_replayed_ragged_epochs_over_a_pool_equal_eager_steps(var_amd, tmp_path):
    """The reference's iTHOR default is 500 triplets at batch 128 (128 / 128 / 128 / 116, Envs/ai2thor/config.py:24,41);
    here the same shape in small: 22 triplets at batch 8 (8 / 8 / 6), clips of up to 1.5 s with ragged lengths, two epochs
    through IthorTrainer.capture_epoch_steps (gather of the row from the resident pool, then the captured step -- a second
    graph for the short batch) against eager step_from_pcm calls on the rows.  Run twice in one
    process."""
    pool = var_amd.SyntheticTripletPool(22, hw=96, seed=5, clips_per_class=3, n_samples=24000, ragged_lens=True).freeze_pairs()
    B, spe, bt = 8, pool.steps_per_epoch(8), pool.tail_batch(8)
    assert (spe, bt) == (3, 6)
    table = pool.index_table(B, 2 * spe, drop_last=False)
    torch.manual_seed(977)
    sd = IthorNetCPU().state_dict()
    out = []
    for mode in ("replay", "eager", "replay"):
        m = var_amd.IthorVARPretextNet(icfg(96))
        m.load_state_dict(sd)
        m = m.to("cuda")
        tr = var_amd.IthorTrainer(m, lr=1e-3)
        losses = []
        if mode == "replay":
            replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table, steps_per_epoch=spe, tail_batch=bt)
            assert torch.equal(m.flat_parameters().cpu(), torch.cat([v.reshape(-1) for v in sd.values()]))   # capture trains nothing
            losses = [float(replay().item()) for _ in range(2 * spe)]
        else:
            for row in range(2 * spe):
                Bs = bt if row % spe == spe - 1 else B
                r = table[row]
                img = pool.images[r[:Bs].long()].contiguous()
                pcm = pool.clips[r[Bs:3 * Bs].long()].contiguous()
                losses.append(float(tr.step_from_pcm(img, pcm, r[3 * Bs:5 * Bs].contiguous()).item()))
        out.append((losses, m.flat_parameters().clone(), tr.step_count))
    assert out[0][2] == out[1][2] == out[2][2] == 6
    assert out[0][0] == out[2][0] and torch.equal(out[0][1], out[2][1])
    assert np.allclose(out[0][0], out[1][0], rtol=0, atol=2e-6), (out[0][0], out[1][0])
    d = (out[0][1] - out[1][1]).abs()
    assert float((d < 2e-6).float().mean()) > 0.995 and float(d.max()) < 5e-3      # (host-side vs device-side Adam scalars)
    # and the training loop of VAR/pretext_VAR.py over the pool, with the model's own milestones
    m = var_amd.IthorVARPretextNet(icfg(96)).to("cuda")
    losses = var_amd.train_representation_from_pool(m, pool, epochs=2, batch=8, lr=1e-3, save_dir=str(tmp_path), save_interval=1,
                                                    log=lambda *a: None)
    assert len(losses) == 2 and all(np.isfinite(v) for v in losses)
    assert sorted(p.name for p in tmp_path.iterdir()) == ['0.pt', '1.pt', 'progress.csv']


def test_ithor_replays_survive_synchronises_between_them(var_amd):
    """Regression: with memset nodes in the captured step (hipMemsetAsync for the GRU's initial state, the gradient arena, ...)
    a replay after a device or stream synchronise filled those buffers with garbage in about half of the processes (loss ==
    margin, zero gradients).  The library now zeroes by kernel.  Zero learning rate: every replay of the
    same batch must return the same loss and the same gradient, whatever is synchronised in between."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    pool = var_amd.SyntheticTripletPool(22, hw=96, seed=5, clips_per_class=3, n_samples=24000, ragged_lens=True).freeze_pairs()
    B = 8
    row = pool.index_table(B, 1, drop_last=True)[0]
    img, pcm, lens = pool.images[row[:B].long()].contiguous(), pool.clips[row[B:3 * B].long()].contiguous(), row[3 * B:5 * B].contiguous()
    torch.manual_seed(977)
    m = var_amd.IthorVARPretextNet(icfg(96)).to("cuda")
    tr = var_amd.IthorTrainer(m, lr=0.0)
    replay = tr.capture_step(img, pcm, lens, _ctx=tr.ctx)
    l0 = float(replay().item())
    g0 = tr.grads.clone()
    assert np.isfinite(l0) and abs(l0 - 1.0) > 1e-3 and float(g0.abs().max()) > 1e-3
    for sync in (lambda: hip.hipStreamSynchronize(None), torch.cuda.synchronize, lambda: torch.cuda.default_stream().synchronize()):
        sync()
        for _ in range(2):
            assert float(replay().item()) == l0
            assert torch.equal(tr.grads, g0)


def test_one_launch_image_forward_equals_the_two_launch_forward(var_amd, golden_dir):
    """Round 4: an image-only forward at 84 x 84 (the frozen encoder at a full batch, the projection of a dataset) runs conv 1-5
    and the image head as ONE launch (img_fwd_all_kernel: img_head2's body, then img_mid3's in the same workgroup); a forward
    with a sound branch keeps the two launches (the step is CU-time bound: csrc/api.hip).  Same arithmetic in both: the
    embeddings and the raw features must be bit-identical, at 3 images and at a full batch."""
    sd = load(golden_dir, "kuka_weights.npz")
    m = make_model(var_amd, sd, 84)
    rng = np.random.default_rng(11)
    for B in (3, 256):
        img = torch.from_numpy(rng.integers(0, 256, size=(B, 3, 84, 84), dtype=np.uint8)).cuda()
        snd = torch.from_numpy(rng.standard_normal((B, 1, 100, 40)).astype(np.float32)).cuda()
        with torch.no_grad():
            only = m(img, None, None)                              # image-only: the fused launch
            both = m(img, snd, snd)                                # with a sound branch: two launches
        for k in ("image_feat", "image_feat_raw"):
            assert torch.equal(only[k], both[k]), (B, k)
