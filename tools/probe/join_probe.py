"""Where do the device-side stream hand-overs of a training step time out?  Eager steps, then replayed ones, printing
var_join_status after each."""
import os, sys, types, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import Context
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(1)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m, lr=1e-3)
pool = var_amd.SyntheticTripletPool(4 * B, hw=84, seed=5, clips_per_class=4).freeze_pairs()
ctx = Context.get(0)
table = pool.index_table(B, 4)[:4].contiguous()
print("start", ctx.join_timeouts(), flush=True)
for s in range(4):
    r = table[s]
    t0 = time.perf_counter()
    l = float(tr.step_from_dataset(pool.images, r[:B], pool.clips, r[B:3 * B], r[3 * B:]).item())
    torch.cuda.synchronize()
    print("eager", s, "loss %.5f" % l, "ms %.2f" % (1e3 * (time.perf_counter() - t0)), "timeouts", ctx.join_timeouts(), flush=True)
replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
print("captured", ctx.join_timeouts(), flush=True)
for s in range(6):
    t0 = time.perf_counter()
    l = float(replay().item())
    torch.cuda.synchronize()
    print("replay", s, "loss %.5f" % l, "ms %.2f" % (1e3 * (time.perf_counter() - t0)), "timeouts", ctx.join_timeouts(), flush=True)
