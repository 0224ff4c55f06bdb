"""Two REAL ranks on the GPU box (-m gpu): both processes drive the HIP library on cuda:0 and exchange through
torch.distributed's gloo backend (RCCL refuses two ranks on one device; the collective is not what is under test --
the 1-GPU box has no second device).  What runs here with world_size 2 on real kernels: VARTrainer's parameter
broadcast, the three-graph replay pipeline (gradient graph -> asynchronous all-reduce || next step's MFCC graph -> Adam
graph with the double index row), the ragged last batch of an epoch, and step_inbatch's gather / offset arithmetic --
against ONE process that takes the whole global batch."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _cfg():
    return types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_ROWS = [list(range(0, 64)), list(range(64, 128)), list(range(128, 168)),          # epoch: 64 / 64 / 40 (short last batch)
         list(range(167, 103, -1)), list(range(103, 39, -1)), list(range(39, -1, -1))]


def _tables(pool, world, Bl, rows_global):
    tabs = []
    for r in range(world):
        rows = []
        for items in rows_global:
            n = len(items) // world
            mine = torch.tensor(items[r * n:(r + 1) * n], device=pool.device)
            row = torch.zeros(5 * Bl, dtype=torch.int32, device=pool.device)
            row[:n] = mine.to(torch.int32)
            row[n:3 * n] = pool.clip_tab[:, mine].reshape(-1)
            row[3 * n:5 * n] = pool.len_tab[:, mine].reshape(-1)
            rows.append(row)
        tabs.append(torch.stack(rows).contiguous())
    return tabs


def _worker(rank, world, port, out, mode):
    import torch.distributed as dist
    import var_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7 + rank)                                  # different initial weights: the trainer broadcasts rank 0's
    model = var_amd.VARPretextNet(_cfg()).to("cuda")
    tr = var_amd.VARTrainer(model, lr=1e-3)
    pool = var_amd.SyntheticTripletPool(168, hw=84, seed=3, clips_per_class=4).freeze_pairs()
    if mode == "replay":
        table = _tables(pool, world, 32, _ROWS)[rank]
        replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, 32, table, global_batch=64, steps_per_epoch=3,
                                           tail_batch=20, tail_global_batch=40)
        losses = [float(replay().item()) for _ in range(len(_ROWS))]
    else:                                                        # in-batch negatives, eager
        losses = []
        for items in _ROWS[:2]:
            n = len(items) // world
            mine = torch.tensor(items[rank * n:(rank + 1) * n], device="cuda")
            f = var_amd.mfcc(pool.clips, pool.len_tab[:, mine].reshape(-1), 100, pool.clip_tab[:, mine].reshape(-1))
            losses.append(float(tr.step_inbatch(pool.images[mine].contiguous(), f[:n].contiguous(), f[n:].contiguous(), tau=0.1).item()))
    out.put((rank, losses, model.flat_parameters().cpu().numpy().copy()))
    dist.destroy_process_group()


def _run(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_ranks_replayed_epoch_equals_one_rank_on_the_global_batch():
    import var_amd
    got = _run("replay")
    torch.manual_seed(7)
    model = var_amd.VARPretextNet(_cfg()).to("cuda")
    tr = var_amd.VARTrainer(model, lr=1e-3)
    pool = var_amd.SyntheticTripletPool(168, hw=84, seed=3, clips_per_class=4).freeze_pairs()
    want = []
    for items in _ROWS:
        i = torch.tensor(items, device="cuda")
        want.append(float(tr.step_from_dataset(pool.images, i.to(torch.int32), pool.clips, pool.clip_tab[:, i].reshape(-1).contiguous(),
                                               pool.len_tab[:, i].reshape(-1).contiguous()).item()))
    flat = model.flat_parameters().cpu().numpy()
    for _rank, losses, p in got:
        np.testing.assert_allclose(losses, want, atol=2e-6)
        d = np.abs(p - flat)
        assert d.max() < 1e-4 and np.mean(d < 2e-6) > 0.99, (d.max(), np.mean(d < 2e-6))
    assert np.array_equal(got[0][2], got[1][2])                    # the replicas stay bit-identical


def test_two_ranks_step_inbatch_equals_one_rank_on_the_global_batch():
    import var_amd
    got = _run("inbatch")
    torch.manual_seed(7)
    model = var_amd.VARPretextNet(_cfg()).to("cuda")
    tr = var_amd.VARTrainer(model, lr=1e-3)
    pool = var_amd.SyntheticTripletPool(168, hw=84, seed=3, clips_per_class=4).freeze_pairs()
    want = []
    for items in _ROWS[:2]:
        # one rank scoring all 64 anchors against [p0 ; n0 ; p1 ; n1]: the candidate ORDER of the two-rank run
        n = len(items) // 2
        i = torch.tensor(items, device="cuda")
        f = var_amd.mfcc(pool.clips, pool.len_tab[:, i].reshape(-1), 100, pool.clip_tab[:, i].reshape(-1))
        pos, neg = f[:2 * n], f[2 * n:]
        c, m = tr.ctx, model
        from var_amd._lib import ptr
        emb = torch.empty((3, 2 * n, 3), device="cuda")
        img = pool.images[i].contiguous()
        tr._bind()
        c.ensure_plan(2 * n, 84)
        c.check(c.lib.var_arm_encoder_fwd(c.handle, c.stream(), ptr(m.flat_parameters()), ptr(img), 1, img.stride(0), ptr(pos), ptr(neg),
                                          2 * n, 84, ptr(emb[0]), ptr(emb[1]), ptr(emb[2]), None, None, 1), "fwd")
        cand = torch.cat([emb[1, :n], emb[2, :n], emb[1, n:], emb[2, n:]]).contiguous()
        target = torch.cat([torch.arange(n), torch.arange(n) + 2 * n]).to(torch.int32).cuda()
        loss, ga, gc = var_amd.inbatch_contrastive_loss(emb[0], cand, target, tau=0.1)
        gp = torch.cat([gc[:n], gc[2 * n:3 * n]]).contiguous()
        gn = torch.cat([gc[n:2 * n], gc[3 * n:]]).contiguous()
        c.check(c.lib.var_arm_encoder_bwd(c.handle, c.stream(), ptr(m.flat_parameters()), ptr(ga), ptr(gp), ptr(gn), ptr(tr.gbuf)), "bwd")
        tr.gbuf[var_amd.N_PARAMS:].copy_(loss)
        tr.adam()
        want.append(float(loss.item()))
    flat = model.flat_parameters().cpu().numpy()
    for _rank, losses, p in got:
        np.testing.assert_allclose(losses, want, atol=1e-5)
        d = np.abs(p - flat)
        assert d.max() < 1e-4 and np.mean(d < 2e-6) > 0.99, (d.max(), np.mean(d < 2e-6))
    assert np.array_equal(got[0][2], got[1][2])
