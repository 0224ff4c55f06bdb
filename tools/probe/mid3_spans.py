"""Where a one-workgroup-per-CU kernel's time goes OUTSIDE its workgroups: first / last instruction of every workgroup of
img_mid3_kernel in chip-wide s_memrealtime ticks (phase build: VAR_HIP_LIB=.../libvar_ph_t0.so), against the kernel's event duration."""
import ctypes, os, sys, types, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import load_library, Context
B = 256
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m)
pool = var_amd.SyntheticTripletPool(1024, hw=84, seed=0, clips_per_class=8).freeze_pairs()
tr.ctx.set_streams(0)
def step():
    i, c, l = pool.next_batch_indices(B)
    tr.step_from_dataset(pool.images, i, pool.clips, c, l)
for _ in range(5): step()
torch.cuda.synchronize()
lib = load_library()
buf = (ctypes.c_ulonglong * (2 * B))()
ctx = Context.get(0)
ctx.profile_select(2)
for it in range(4):
    step(); torch.cuda.synchronize()
    lib.var_debug_spans_mid3(buf, B)
    a = np.array(buf[:], dtype=np.int64).reshape(B, 2) / 100.0      # us
    t0 = a[:, 0].min()
    st, en = a[:, 0] - t0, a[:, 1] - t0
    print("starts: min 0 median %.2f max %.2f us | ends: min %.2f median %.2f max %.2f us | span median %.2f max %.2f | slowest blocks %s"
          % (np.median(st), st.max(), en.min(), np.median(en), en.max(), np.median(en - st), (en - st).max(), np.argsort(en)[-4:].tolist()))
ms, n = ctx.profile_read()
print("event duration per launch: %.2f us" % (1e3 * ms / n))
