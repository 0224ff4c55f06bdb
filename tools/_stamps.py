"""Diagnostic: in-kernel s_memtime stamps of the fused backward tail (VAR_STAMPS=1)."""
import os, sys, types
os.environ["VAR_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import var_amd
from var_amd._lib import Context
B = 256
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
model = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(model)
pool = var_amd.SyntheticTripletPool(2048, hw=84, seed=0, clips_per_class=32).freeze_pairs()
ctx = Context.get(0)
for _ in range(3):
    i, c, l = pool.next_batch_indices(B)
    tr.step_from_dataset(pool.images, i, pool.clips, c, l)
torch.cuda.synchronize()
buf = ctx.debug_buffer("slabs")
raw = buf[-512:].cpu().numpy().view(np.int64)
for name, off in (("wave0", 0), ("waveL", 64)):
    st = raw[off:off + 64]
    st = st[st != 0]
    d = (st - st[0]).tolist()
    print(name, d)
    print("  deltas", np.diff(st).tolist())
