"""Soak of the replayed training step's device-side stream hand-over: N replays, then var_join_status and the slowest replays
(a time-out shows as a 5-ms replay).  usage: join_soak.py [replays] [84 | 96]"""
import os, sys, types, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import Context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = 256
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 84
cfg = types.SimpleNamespace(img_dim=(3, HW, HW), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(1)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m, lr=1e-4)
pool = var_amd.SyntheticTripletPool(8192, hw=HW, seed=5, clips_per_class=8).freeze_pairs()
table = pool.index_table(B, 16)[:16].contiguous()
ctx = Context.get(0)
replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
for _ in range(50): replay()
torch.cuda.synchronize()
CH = 500
times = []
acc = torch.zeros(1, device="cuda")
for c in range(N // CH):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(CH):
        acc += replay().reshape(-1)[:1]
    e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / CH)
print("replays", N, "ms/step per chunk of %d: min %.4f median %.4f max %.4f" % (CH, min(times), float(np.median(times)), max(times)))
print("join timeouts", ctx.join_timeouts(), "mean loss %.5f finite %s" % (float(acc.item()) / N, bool(torch.isfinite(acc).item())))
