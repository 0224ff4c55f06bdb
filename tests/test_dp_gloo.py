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
