"""Data-parallel convention of VARTrainer with world_size 2 on gloo/CPU: each rank
computes loss*B_local/B_global and its gradients (inv_count = 1/B_global), ONE all_reduce(SUM) over the
flat arena (+ loss slot) must reproduce the single-process full-batch result.  The compute here is the
CPU oracle (the HIP kernels need a GPU); the sharding, scaling and the arena+loss message are what is tested."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import var_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, golden_dir, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights2.npz")))
    fx = dict(np.load(os.path.join(golden_dir, "kuka_h84_w2.npz")))
    P = orc.flatten_params(sd)
    B = fx['image'].shape[0]                       # 7: unequal shards 4 + 3
    lo, hi = (0, 4) if rank == 0 else (4, B)
    loss, G, _ = orc.loss_grad(P, fx['image'][lo:hi], fx['sound_positive'][lo:hi], fx['sound_negative'][lo:hi])
    w = (hi - lo) / B                              # mean over the shard -> share of the global mean
    buf = torch.from_numpy(np.concatenate([G * w, [loss * w]]).astype(np.float32))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)     # the ONE collective of the step
    if rank == 0:
        out.put(buf.numpy())
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_full_batch(golden_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights2.npz")))
    fx = dict(np.load(os.path.join(golden_dir, "kuka_h84_w2.npz")))
    loss, G, _ = orc.loss_grad(orc.flatten_params(sd), fx['image'], fx['sound_positive'], fx['sound_negative'])
    assert abs(got[-1] - loss) < 1e-6
    assert abs(got[-1] - float(fx['loss'])) < 1e-5
    assert np.max(np.abs(got[:-1] - G)) < 1e-6 * max(1.0, np.max(np.abs(G)))


def _ithor_worker(rank, world, port, golden_dir, out):
    from oracle.torch_oracle import ithor_seeded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = dict(np.load(os.path.join(golden_dir, "ithor_h96.npz")))
    m = ithor_seeded(int(fx["seed"]))
    B = fx["image"].shape[0]                       # 2: one triplet per rank
    sl = slice(rank, rank + 1)
    a, p, n = m((torch.from_numpy(fx["image"][sl]) / 255.).float(), torch.from_numpy(fx["sound_positive"][sl]),
                torch.from_numpy(fx["sound_negative"][sl]))
    loss = torch.nn.TripletMarginLoss(margin=1.0, p=2, reduction="sum")(a, p, n) / B     # inv_count = 1/B_global
    loss.backward()
    buf = torch.cat([q.grad.reshape(-1) for q in m.parameters()] + [loss.detach().reshape(1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)     # gradient arena + loss slot, as IthorTrainer.allreduce
    if rank == 0:
        out.put(buf.numpy())
    dist.destroy_process_group()


def test_ithor_two_rank_allreduce_equals_reference_full_batch(golden_dir):
    """The same convention for the iTHOR model (BASELINE config 4): two ranks, one triplet each, against the
    reference's full-batch loss and gradient samples of tests/golden/ithor_h96.npz."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ithor_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fx = dict(np.load(os.path.join(golden_dir, "ithor_h96.npz")))
    assert abs(got[-1] - float(fx["losses"][0])) < 1e-6
    from oracle.torch_oracle import ithor_seeded
    stride, o = int(fx["stride"]), 0
    for k, p in ithor_seeded(int(fx["seed"])).named_parameters():
        g = got[o:o + p.numel()]
        o += p.numel()
        np.testing.assert_allclose(g[::stride], fx["gsamp." + k], rtol=1e-4, atol=1e-7, err_msg=k)


def _inbatch_worker(rank, world, port, out):
    """The data-parallel schedule of VARTrainer.step_inbatch on gloo/CPU with the torch restatement of the loss:
    all-gather of [pos ; neg], local rows vs all candidates, all-reduce of the candidate gradients, own slice."""
    from oracle.torch_oracle import inbatch_contrastive_loss
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100)
    Bl = 5
    emb = torch.nn.functional.normalize(torch.randn(world, 3, Bl, 3, generator=g), dim=-1)    # every rank's [a | p | n]
    a = emb[rank, 0].clone().requires_grad_()
    local = emb[rank, 1:].reshape(2 * Bl, 3).clone()
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    cand = torch.cat(parts).requires_grad_()
    target = torch.arange(Bl) + rank * 2 * Bl
    loss = inbatch_contrastive_loss(a, cand, target, tau=0.1, inv_count=1.0 / (Bl * world))
    loss.backward()
    gc = cand.grad.clone()
    dist.all_reduce(gc, op=dist.ReduceOp.SUM)
    lsum = loss.detach().clone().reshape(1)
    dist.all_reduce(lsum, op=dist.ReduceOp.SUM)
    out.put((rank, lsum.item(), a.grad.numpy().copy(), gc[rank * 2 * Bl:(rank + 1) * 2 * Bl].numpy().copy()))
    dist.destroy_process_group()


def test_inbatch_negatives_two_rank_schedule_equals_full_batch():
    from oracle.torch_oracle import inbatch_contrastive_loss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inbatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(100)
    Bl, world = 5, 2
    emb = torch.nn.functional.normalize(torch.randn(world, 3, Bl, 3, generator=g), dim=-1)
    a = emb[:, 0].reshape(world * Bl, 3).clone().requires_grad_()
    cand = emb[:, 1:].reshape(world * 2 * Bl, 3).clone().requires_grad_()          # [p0 ; n0 ; p1 ; n1]
    target = torch.cat([torch.arange(Bl) + r * 2 * Bl for r in range(world)])
    full = inbatch_contrastive_loss(a, cand, target, tau=0.1)
    full.backward()
    for r, lsum, ga, gmine in got:
        assert abs(lsum - full.item()) < 1e-6
        np.testing.assert_allclose(ga, a.grad[r * Bl:(r + 1) * Bl].numpy(), atol=1e-6)
        np.testing.assert_allclose(gmine, cand.grad[r * 2 * Bl:(r + 1) * 2 * Bl].numpy(), atol=1e-6)
