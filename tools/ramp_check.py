"""Per-replay device time of the captured Kuka step right after capture (how long the first replays take to settle)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, var_amd
B = 256
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m, lr=float(os.environ.get("RAMP_LR", "1e-4")))
pool = var_amd.SyntheticTripletPool(4096, hw=84, seed=0, clips_per_class=32).freeze_pairs()
for _ in range(60):
    i, c, l = pool.next_batch_indices(B)
    tr.step_from_dataset(pool.images, i, pool.clips, c, l)
torch.cuda.synchronize()
t0 = time.perf_counter()
replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, pool.index_table(B, 256)[:256].contiguous())
torch.cuda.synchronize()
print("capture took %.1f ms" % (1e3 * (time.perf_counter() - t0)))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode.startswith("busy"):  # keep the GPU busy with eager steps right before the replays ("busy_sync": one step in flight)
    for _ in range(60):
        i, c, l = pool.next_batch_indices(B)
        tr.step_from_dataset(pool.images, i, pool.clips, c, l)
        if mode == "busy_sync":
            torch.cuda.synchronize()
n = 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for k in range(n):
    replay()
    ev[k + 1].record()
torch.cuda.synchronize()
d = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
print(mode, " ".join("%.3f" % x for x in d[:40]))
print("mean 0-4 %.4f  5-24 %.4f  25-79 %.4f" % (sum(d[:5]) / 5, sum(d[5:25]) / 20, sum(d[25:]) / 55))
