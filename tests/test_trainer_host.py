"""Host logic of VARTrainer / TripletPool on the CPU: the trainer's own code (index-row layouts, ragged last batch,
replay bookkeeping, device-side cursor protocol) driven through tests/_oracle_ctx.py, against the torch restatement of
the reference step (oracle/torch_oracle.py:CPUTrainer) fed the same batches.  No HIP compute here -- the GPU suite
(tests/test_gpu_parity.py) runs the same scenarios on the real library."""
import types

import numpy as np
import pytest
import torch

import var_amd
from oracle import mfcc_np
from oracle.torch_oracle import CPUTrainer
from tests._oracle_ctx import OracleContext


def _cfg(hw=84):
    return types.SimpleNamespace(img_dim=(3, hw, hw), sound_dim=(1, 100, 40), representationDim=3)


def _pool(n_items, seed=3):
    # 2 clips per class keeps the numpy MFCC of the oracle context cheap
    return var_amd.SyntheticTripletPool(n_items, hw=84, seed=seed, clips_per_class=2, device="cpu").freeze_pairs()


def _features(pool, clip_idx, lens):
    out = np.zeros((len(clip_idx), 1, 100, 40), np.float32)
    clips = pool.clips.numpy()
    for i, (c, n) in enumerate(zip(clip_idx.tolist(), lens.tolist())):
        if n > 0:
            out[i] = mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(clips[c, :n], dtype=np.float32))
    return torch.from_numpy(out)


def test_epoch_table_keeps_the_short_last_batch():
    """DataLoader(drop_last=False) of the reference (dataset.py:157-162): 300 triplets at batch 128 are 3 steps, 128 /
    128 / 44; every item appears exactly once per epoch."""
    pool = var_amd.SyntheticTripletPool(300, hw=84, seed=1, clips_per_class=2, device="cpu").freeze_pairs()
    assert pool.steps_per_epoch(128) == 3 and pool.tail_batch(128) == 44
    assert pool.steps_per_epoch(128, drop_last=True) == 2 and pool.tail_batch(128, drop_last=True) == 0
    tab = pool.epoch_index_table(128)
    assert tab.shape == (3, 640) and tab.dtype == torch.int32
    seen = torch.cat([tab[0, :128], tab[1, :128], tab[2, :44]]).long()
    assert sorted(seen.tolist()) == list(range(300))
    # the short row is packed [image (44) | clips (88) | lens (88)] and zero beyond
    idx = tab[2, :44].long()
    assert torch.equal(tab[2, 44:88], pool.clip_tab[0, idx]) and torch.equal(tab[2, 88:132], pool.clip_tab[1, idx])
    assert torch.equal(tab[2, 132:176], pool.len_tab[0, idx]) and torch.equal(tab[2, 176:220], pool.len_tab[1, idx])
    assert int(tab[2, 220:].abs().sum()) == 0
    assert pool.epoch_index_table(128, drop_last=True).shape == (2, 640)
    assert pool.index_table(128, 7, drop_last=False).shape[0] == 9     # whole epochs, each with its short row
    assert pool.index_table(128, 7).shape[0] == 8                      # default: full rows only (a padded row run as a full batch would train on zeros)


def test_replayed_ragged_epochs_equal_reference_steps():
    """Two epochs of 22 triplets at batch 8 (8 / 8 / 6) through capture_epoch_steps(tail_batch=6): per-step losses and
    the final parameters equal the reference step (torch restatement) on the same batches, the short batch averaged
    over ITS size as TripletMarginLoss(reduction='mean') does."""
    torch.manual_seed(11)
    model = var_amd.VARPretextNet(_cfg())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ctx = OracleContext()
    tr = var_amd.VARTrainer(model, lr=1e-3, _ctx=ctx)
    pool = _pool(22)
    B, spe, bt = 8, pool.steps_per_epoch(8), pool.tail_batch(8)
    assert (spe, bt) == (3, 6)
    table = pool.index_table(B, 2 * spe, drop_last=False)
    replay, _load = tr.capture_epoch_steps(pool.images, pool.clips, B, table, steps_per_epoch=spe, tail_batch=bt)
    ref = CPUTrainer(state_dict=sd, lr=1e-3)
    for row in range(2 * spe):
        Bs = bt if row % spe == spe - 1 else B
        r = table[row]
        want = ref.step(pool.images[r[:Bs].long()], _features(pool, r[Bs:2 * Bs], r[3 * Bs:4 * Bs]),
                        _features(pool, r[2 * Bs:3 * Bs], r[4 * Bs:5 * Bs]))
        got = float(replay().item())
        assert abs(got - want) < 2e-6, (row, got, want)
    sizes = [c[1] for c in ctx.lib.calls if c[0] == "loss_grad_pcm"]
    assert sizes == [8, 8, 6, 8, 8, 6]
    assert [round(1 / c[2]) for c in ctx.lib.calls if c[0] == "loss_grad_pcm"] == sizes      # mean over each batch
    flat_ref = torch.cat([ref.model.state_dict()[k].reshape(-1) for k, _ in var_amd.PARAM_SPECS])
    assert float((model.flat_parameters() - flat_ref).abs().max()) < 2e-6
    assert tr.step_count == 6
    # a 7th replay wraps to row 0 of the table (full batch again)
    replay()
    assert [c[1] for c in ctx.lib.calls if c[0] == "loss_grad_pcm"][-1] == 8


def test_ragged_table_arguments_are_checked():
    model = var_amd.VARPretextNet(_cfg())
    tr = var_amd.VARTrainer(model, _ctx=OracleContext())
    pool = _pool(22)
    table = pool.index_table(8, 3, drop_last=False)
    with pytest.raises(var_amd.VarHipError):
        tr.capture_epoch_steps(pool.images, pool.clips, 8, table, steps_per_epoch=2, tail_batch=6)   # 3 % 2 != 0
    with pytest.raises(var_amd.VarHipError):
        tr.capture_epoch_steps(pool.images, pool.clips, 8, table, steps_per_epoch=3, tail_batch=8)   # tail == batch
    with pytest.raises(var_amd.VarHipError):
        tr.capture_epoch_steps(pool.images, pool.clips, 8, table.long(), steps_per_epoch=3, tail_batch=6)


def test_trainer_without_a_device_context_refuses_the_cpu():
    """The seam above is for tests: the product has no CPU path."""
    with pytest.raises(var_amd.VarHipError):
        var_amd.VARTrainer(var_amd.VARPretextNet(_cfg()))


def test_two_level_clip_draw_and_load_num(tmp_path):
    """Envs/audioLoader.py:174-176 picks a dataset, then a clip: with datasets of 1 and 9 clips per class the single
    clip of the small dataset is drawn about half of the time (a flat draw would give 10 %).  dataset.py:150-154:
    loadNum picks a random subset of the pickles."""
    import pickle
    task_num, n = 4, 4000
    clips = np.zeros((task_num * 10, 16), np.int16)
    pool = var_amd.TripletPool(np.zeros((n, 3, 84, 84), np.uint8), np.zeros(n, np.int64), np.ones(n, np.int64), clips,
                               np.full(40, 16, np.int32), np.arange(task_num) * 10, np.full(task_num, 10), task_num,
                               device="cpu")
    ds_start = np.stack([np.arange(task_num) * 10, np.arange(task_num) * 10 + 1], axis=1)     # dataset 0: 1 clip, dataset 1: 9
    pool.set_datasets(ds_start, np.tile([1, 9], (task_num, 1))).freeze_pairs()
    pos = pool.clip_tab[0]
    assert int(pos.min()) >= 0 and int(pos.max()) <= 9                    # class 0 only
    frac = float((pos == 0).float().mean())
    assert 0.45 < frac < 0.55, frac
    # a class missing from one dataset never draws from it
    pool.set_datasets(ds_start, np.tile([0, 9], (task_num, 1))).freeze_pairs()
    assert int((pool.clip_tab[0] == 0).sum()) == 0
    paths = []
    for k in range(5):
        p = tmp_path / f"data_{k}.pickle"
        with open(p, "wb") as f:
            pickle.dump([{"image": np.full((3, 84, 84), k, np.uint8), "ground_truth": np.array([k % 4]),
                          "sound_negative_id": np.array([(k + 1) % 4])} for _ in range(3)], f)
        paths.append(str(p))
    args = (clips, np.full(40, 16, np.int32), np.arange(task_num) * 10, np.full(task_num, 10), task_num)
    sub = var_amd.TripletPool.from_pickles(paths, *args, device="cpu", load_num=2)
    assert sub.n_items == 6 and len(set(sub.images[:, 0, 0, 0].tolist())) == 2
    assert var_amd.TripletPool.from_pickles(paths, *args, device="cpu", load_num='all').n_items == 15


def test_replayed_inbatch_step_equals_eager_step_inbatch():
    """capture_inbatch_epoch_steps (gather + MFCC + encoder + in-batch head + backward + Adam + row fetch as a replayed
    step) walks the index table like eager step_inbatch calls on the same rows fed the same features."""
    torch.manual_seed(3)
    ma, mb = var_amd.VARPretextNet(_cfg()), var_amd.VARPretextNet(_cfg())
    mb.load_state_dict(ma.state_dict())
    ta = var_amd.VARTrainer(ma, lr=1e-3, _ctx=OracleContext())
    tb = var_amd.VARTrainer(mb, lr=1e-3, _ctx=OracleContext())
    pool = _pool(12)
    B = 4
    table = pool.index_table(B, 3, drop_last=True)
    replay, _ = tb.capture_inbatch_epoch_steps(pool.images, pool.clips, B, table, tau=0.1)
    for s in range(4):                                             # the 4th replay wraps to row 0
        r = table[s % 3]
        want = float(ta.step_inbatch(pool.images[r[:B].long()].contiguous(), _features(pool, r[B:2 * B], r[3 * B:4 * B]),
                                     _features(pool, r[2 * B:3 * B], r[4 * B:]), tau=0.1).item())
        got = float(replay().item())
        assert abs(got - want) < 1e-6, (s, got, want)
    assert float((ma.flat_parameters() - mb.flat_parameters()).abs().max()) < 2e-5      # 4 Adam steps of lr 1e-3


def test_pool_training_loop_equals_the_reference_loop(tmp_path):
    """train_representation_from_pool (shuffled epoch tables, ragged last batch, per-epoch MultiStepLR, device-side loss
    accumulation, legacy .pt + progress.csv) against the reference's loop restated on the CPU over the same epoch
    tables: per-epoch average losses and final weights."""
    import csv
    torch.manual_seed(21)
    model = var_amd.VARPretextNet(_cfg())
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    pool = _pool(14, seed=5)
    pool_ref = _pool(14, seed=5)                                   # same draws: the reference loop's tables
    B, epochs, milestones = 4, 3, (1, 2)
    out = var_amd.train_representation_from_pool(model, pool, epochs, B, lr=1e-3, milestones=milestones, gamma=0.5,
                                                 save_dir=str(tmp_path), save_interval=2, log=lambda *a: None,
                                                 _ctx=OracleContext())
    ref = CPUTrainer(state_dict=sd, lr=1e-3)
    want = []
    spe, bt = pool_ref.steps_per_epoch(B), pool_ref.tail_batch(B)
    assert (spe, bt) == (4, 2)
    for ep in range(epochs):
        for g in ref.opt.param_groups:
            g['lr'] = var_amd.multistep_lr(1e-3, milestones, 0.5, ep)
        table = pool_ref.epoch_index_table(B)
        losses = []
        for row in range(spe):
            Bs = bt if row == spe - 1 else B
            r = table[row]
            losses.append(ref.step(pool_ref.images[r[:Bs].long()], _features(pool_ref, r[Bs:2 * Bs], r[3 * Bs:4 * Bs]),
                                   _features(pool_ref, r[2 * Bs:3 * Bs], r[4 * Bs:5 * Bs])))
        want.append(sum(losses) / len(losses))
    np.testing.assert_allclose(out, want, atol=3e-6)
    flat_ref = torch.cat([ref.model.state_dict()[k].reshape(-1) for k, _ in var_amd.PARAM_SPECS])
    assert float((model.flat_parameters() - flat_ref).abs().max()) < 3e-5
    assert sorted(p.name for p in tmp_path.iterdir()) == ['1.pt', '2.pt', 'progress.csv']
    with open(tmp_path / 'progress.csv') as f:
        rows = list(csv.reader(f))
    assert rows[0] == ['avg_loss'] and len(rows) == 1 + epochs
    loaded = torch.load(tmp_path / '2.pt', weights_only=True)
    assert list(loaded.keys()) == [k for k, _ in var_amd.PARAM_SPECS]


def _bench_dry_run(argv):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable] + argv, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                           # ONE JSON line, from rank 0, whatever the other ranks print
    return json.loads(lines[0])


def test_bench_launch_logic_with_eight_ranks_on_the_cpu():
    """No 8-GPU node was available to the driver in rounds 1-3, so the N > 1 launch path of bench.py never ran at its real
    width.  --dry-run-cpu runs everything of it that is not a kernel: the driver's launch line (torch.distributed.run,
    --nproc-per-node 8, 127.0.0.1 rendezvous), RANK / WORLD_SIZE handling, the process group, the barriers around the timed
    region, MAX over ranks, one JSON line from rank 0 with whole-job values -- and the self-launching form
    `python bench.py --gpus N` (a child torch.distributed.run, started before anything touches a GPU)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = _bench_dry_run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "bench.py", "--gpus", "8", "--steps", "4", "--warmup", "1", "--dry-run-cpu"])
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8 and out["steps"] == 4 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 8 * out["config"]["per_gpu_batch"] and out["scaling"] == "weak"
    assert abs(out["value"] - 4 * out["config"]["global_batch"] / (4 * out["ms_per_step"] * 1e-3)) < 1e-3 * out["value"]
    assert "dry_run" in out
    out = _bench_dry_run(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu"])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2
