"""Data parallelism of VARTrainer with world_size 2 on gloo/CPU.  The code under test is
VARTrainer's OWN: parameter broadcast at construction, shard scaling (inv_count = 1/B_global), the [gradients | loss]
message, the three-graph replay pipeline with the asynchronous all-reduce and the front-end running one row ahead, the
ragged last batch, and step_inbatch's all-gather / rank-offset / candidate-gradient arithmetic.  Its C-ABI calls are
bound to the CPU oracle (tests/_oracle_ctx.py) -- the HIP kernels need a GPU and are covered by the -m gpu suite."""
import os
import socket
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import var_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _run(worker, *args, world=2, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _cfg():
    return types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)


def _flat_sd(flat):
    return {k: torch.from_numpy(v.copy()) for k, v in orc.unflatten_params(flat.numpy()).items()}


# ---- eager step, unequal shards -------------------------------------------------------------------------------------
def _eager_worker(rank, world, port, out, golden_dir):
    import var_amd
    from tests._oracle_ctx import OracleContext
    _init(rank, world, port)
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights2.npz")))
    fx = dict(np.load(os.path.join(golden_dir, "kuka_h84_w2.npz")))
    torch.manual_seed(1000 + rank)                             # ranks start from DIFFERENT weights ...
    model = var_amd.VARPretextNet(_cfg())
    if rank == 0:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr = var_amd.VARTrainer(model, _ctx=OracleContext())       # ... the constructor broadcasts rank 0's
    assert np.array_equal(model.flat_parameters().numpy(), orc.flatten_params(sd))
    B = fx['image'].shape[0]                                   # 7: unequal shards 4 + 3
    lo, hi = (0, 4) if rank == 0 else (4, B)
    gb = tr.sync_global_batch(hi - lo)
    assert gb == B
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    tr.loss_and_grads(t(fx['image'][lo:hi]), t(fx['sound_positive'][lo:hi]), t(fx['sound_negative'][lo:hi]), global_batch=gb)
    tr.allreduce()                                             # the ONE collective of the step
    out.put((rank, tr.gbuf.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_uneven_shards_allreduce_equals_full_batch(golden_dir):
    got = _run(_eager_worker, golden_dir)
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights2.npz")))
    fx = dict(np.load(os.path.join(golden_dir, "kuka_h84_w2.npz")))
    loss, G, _ = orc.loss_grad(orc.flatten_params(sd), fx['image'], fx['sound_positive'], fx['sound_negative'])
    for _rank, buf in got:
        assert abs(buf[-1] - loss) < 1e-6
        assert abs(buf[-1] - float(fx['loss'])) < 1e-5
        assert np.max(np.abs(buf[:-1] - G)) < 1e-5 * max(1.0, np.max(np.abs(G)))
    assert np.array_equal(got[0][1], got[1][1])                # every rank holds the same reduced arena


# ---- the replayed data-parallel epoch (three graphs, asynchronous all-reduce, ragged last batch) ---------------------
def _dp_tables(pool, world, Bl, rows_global):
    """Per-rank step tables for global batches given as lists of item ids: rank r takes the r-th slice of each batch."""
    tabs = []
    for r in range(world):
        rows = []
        for items in rows_global:
            n = len(items) // world
            mine = torch.tensor(items[r * n:(r + 1) * n])
            row = torch.zeros(5 * Bl, dtype=torch.int32)
            row[:n] = mine.to(torch.int32)
            row[n:3 * n] = pool.clip_tab[:, mine].reshape(-1)
            row[3 * n:5 * n] = pool.len_tab[:, mine].reshape(-1)
            rows.append(row)
        tabs.append(torch.stack(rows).contiguous())
    return tabs


_DP_ROWS = [list(range(0, 8)), list(range(8, 16)), list(range(16, 22)),        # epoch 1: 8 / 8 / 6 (short last batch)
            list(range(21, 13, -1)), list(range(13, 5, -1)), list(range(5, -1, -1))]


def _dp_worker(rank, world, port, out):
    import var_amd
    from tests._oracle_ctx import OracleContext
    _init(rank, world, port)
    torch.manual_seed(7 + rank)
    model = var_amd.VARPretextNet(_cfg())
    tr = var_amd.VARTrainer(model, lr=1e-3, _ctx=OracleContext())
    pool = var_amd.SyntheticTripletPool(22, hw=84, seed=3, clips_per_class=2, device="cpu").freeze_pairs()
    table = _dp_tables(pool, world, 4, _DP_ROWS)[rank]
    replay, _load = tr.capture_epoch_steps(pool.images, pool.clips, 4, table, global_batch=8, steps_per_epoch=3,
                                           tail_batch=3, tail_global_batch=6)
    losses = [float(replay().item()) for _ in range(len(_DP_ROWS))]
    kinds = [c[0] for c in tr.ctx.lib.calls]
    out.put((rank, losses, model.flat_parameters().numpy().copy(), kinds))
    dist.destroy_process_group()


def test_two_rank_replayed_epoch_equals_single_process_reference():
    import var_amd
    from oracle.torch_oracle import CPUTrainer
    from tests.test_trainer_host import _features
    got = _run(_dp_worker)
    torch.manual_seed(7)                                       # rank 0's initial weights (broadcast to rank 1)
    sd = {k: v.detach().clone() for k, v in var_amd.VARPretextNet(_cfg()).state_dict().items()}
    pool = var_amd.SyntheticTripletPool(22, hw=84, seed=3, clips_per_class=2, device="cpu").freeze_pairs()
    ref = CPUTrainer(state_dict=sd, lr=1e-3)
    want = []
    for items in _DP_ROWS:
        i = torch.tensor(items)
        want.append(ref.step(pool.images[i], _features(pool, pool.clip_tab[0, i], pool.len_tab[0, i]),
                             _features(pool, pool.clip_tab[1, i], pool.len_tab[1, i])))
    flat_ref = torch.cat([ref.model.state_dict()[k].reshape(-1) for k, _ in var_amd.PARAM_SPECS]).numpy()
    for _rank, losses, flat, kinds in got:
        np.testing.assert_allclose(losses, want, atol=3e-6)
        # 6 Adam steps of lr 1e-3 can move a parameter by 6e-3; the two summation orders (shard sums added by the
        # all-reduce vs one batch sum) differ in the last bits of the gradients: bound the drift at 0.3 % of that reach
        d = np.abs(flat - flat_ref)
        assert d.max() < 2e-5 and np.mean(d < 1e-6) > 0.999, (d.max(), np.mean(d < 1e-6))
        # the pipeline: features of row 0 up front, then per step gradient pass -> front-end of the NEXT row -> Adam
        assert kinds[:2] == ["mfcc", "mfcc"]                   # warm-up + load_table
        assert kinds[2:] == ["loss_grad_gather", "mfcc", "adam_graph"] * len(_DP_ROWS)
    assert np.array_equal(got[0][2], got[1][2])                # replicas stay bit-identical


def _ithor_worker(rank, world, port, out, golden_dir):
    from oracle.torch_oracle import ithor_seeded
    _init(rank, world, port)
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
    out.put((rank, buf.numpy()))
    dist.destroy_process_group()


def test_ithor_two_rank_allreduce_equals_reference_full_batch(golden_dir):
    """The same convention for the iTHOR model (BASELINE config 4): two ranks, one triplet each, against the
    reference's full-batch loss and gradient samples of tests/golden/ithor_h96.npz."""
    got = _run(_ithor_worker, golden_dir)[0][1]
    fx = dict(np.load(os.path.join(golden_dir, "ithor_h96.npz")))
    assert abs(got[-1] - float(fx["losses"][0])) < 1e-6
    from oracle.torch_oracle import ithor_seeded
    stride, o = int(fx["stride"]), 0
    for k, p in ithor_seeded(int(fx["seed"])).named_parameters():
        g = got[o:o + p.numel()]
        o += p.numel()
        np.testing.assert_allclose(g[::stride], fx["gsamp." + k], rtol=1e-4, atol=1e-7, err_msg=k)


# ---- in-batch negatives (BASELINE configs[2]) through VARTrainer.step_inbatch -------------------------------------------
def _inbatch_worker(rank, world, port, out):
    import var_amd
    from tests._oracle_ctx import OracleContext
    _init(rank, world, port)
    torch.manual_seed(5)
    model = var_amd.VARPretextNet(_cfg())
    tr = var_amd.VARTrainer(model, lr=1e-3, _ctx=OracleContext())
    g = torch.Generator().manual_seed(100)
    Bl = 3
    img = torch.randint(0, 256, (world * Bl, 3, 84, 84), dtype=torch.uint8, generator=g)
    snd = torch.randn(2, world * Bl, 1, 100, 40, generator=g) * 4
    sl = slice(rank * Bl, (rank + 1) * Bl)
    tr.step_inbatch(img[sl].contiguous(), snd[0, sl].contiguous(), snd[1, sl].contiguous(), tau=0.1)
    out.put((rank, tr.gbuf.numpy().copy(), model.flat_parameters().numpy().copy()))
    dist.destroy_process_group()


def test_step_inbatch_two_ranks_equals_full_batch_restatement():
    """VARTrainer.step_inbatch with 2 ranks x 3 triplets against ONE process scoring all 6 anchors against all 12
    candidates [p0 ; n0 ; p1 ; n1] with torch autograd end to end (encoder included)."""
    import var_amd
    from oracle.torch_oracle import KukaNetCPU, inbatch_contrastive_loss
    got = _run(_inbatch_worker)
    torch.manual_seed(5)
    sd = {k: v.detach().clone() for k, v in var_amd.VARPretextNet(_cfg()).state_dict().items()}
    net = KukaNetCPU()
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(100)
    world, Bl = 2, 3
    img = torch.randint(0, 256, (world * Bl, 3, 84, 84), dtype=torch.uint8, generator=g)
    snd = torch.randn(2, world * Bl, 1, 100, 40, generator=g) * 4
    a, p, n = net(img.float() / 255., snd[0], snd[1])
    cand = torch.cat([torch.cat([p[r * Bl:(r + 1) * Bl], n[r * Bl:(r + 1) * Bl]]) for r in range(world)])
    target = torch.cat([torch.arange(Bl) + r * 2 * Bl for r in range(world)])
    loss = inbatch_contrastive_loss(a, cand, target, tau=0.1)
    loss.backward()
    G = torch.cat([dict(net.named_parameters())[k].grad.reshape(-1) for k, _ in var_amd.PARAM_SPECS]).numpy()
    for _rank, buf, _flat in got:
        assert abs(buf[-1] - float(loss)) < 2e-6
        assert np.max(np.abs(buf[:-1] - G)) < 1e-5 * max(1.0, np.max(np.abs(G)))
    assert np.array_equal(got[0][2], got[1][2])
