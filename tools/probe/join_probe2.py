"""The sequence of tests/test_gpu_round4.py, with var_join_status printed after every stage."""
import os, sys, types, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import Context
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
ctx = Context.get(0)
for B in (48, 256):
    pool = var_amd.SyntheticTripletPool(4 * B, hw=84, seed=5, clips_per_class=4).freeze_pairs()
    table = pool.index_table(B, 4)[:4].contiguous()
    for mode, mask in (("flag", 3), ("edge", 3 | 64)):
        old = ctx.set_streams(mask)
        torch.manual_seed(1)
        m = var_amd.VARPretextNet(cfg).to("cuda")
        tr = var_amd.VARTrainer(m, lr=1e-3, weight_decay=1e-6)
        replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
        print(B, mode, "captured", ctx.join_timeouts(), flush=True)
        for s in range(6):
            t0 = time.perf_counter()
            l = float(replay().item())
            torch.cuda.synchronize()
            print(B, mode, "replay", s, "ms %.2f" % (1e3 * (time.perf_counter() - t0)), "timeouts", ctx.join_timeouts(), flush=True)
        r = table[2]
        t0 = time.perf_counter()
        l = float(tr.step_from_dataset(pool.images, r[:B], pool.clips, r[B:3 * B], r[3 * B:]).item())
        torch.cuda.synchronize()
        print(B, mode, "eager", "ms %.2f" % (1e3 * (time.perf_counter() - t0)), "timeouts", ctx.join_timeouts(), flush=True)
        del tr, m
        ctx.set_streams(old)
