"""Start / end of every workgroup of img_head2_kernel inside the REPLAYED two-stream step (phase build:
VAR_HIP_LIB=.../libvar_ph.so): how far the MFCC's workgroups, which hold the CUs' LDS at the start of a step, stagger them."""
import ctypes, os, sys, types, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import load_library
B = 256
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m)
pool = var_amd.SyntheticTripletPool(4096, hw=84, seed=0, clips_per_class=8).freeze_pairs()
table = pool.index_table(B, 8)[:8].contiguous()
replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
for _ in range(20): replay()
torch.cuda.synchronize()
lib = load_library()
buf = (ctypes.c_ulonglong * (2 * B))()
for it in range(4):
    replay(); torch.cuda.synchronize()
    lib.var_debug_spans_head2(buf, B)
    a = np.array(buf[:], dtype=np.int64).reshape(B, 2) / 100.0      # us
    t0 = a[:, 0].min()
    st, en = a[:, 0] - t0, a[:, 1] - t0
    q = np.percentile(st, [10, 25, 50, 75, 90, 100])
    print("starts (us after the first): p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f | ends: min %.1f p50 %.1f max %.1f | own time p50 %.1f max %.1f"
          % (*q, en.min(), np.median(en), en.max(), np.median(en - st), (en - st).max()))
