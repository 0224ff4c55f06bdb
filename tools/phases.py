"""Phase timing of an instrumented build (VAR_HIP_LIB=.../libvar_ph.so): cycles of thread 0 of one workgroup between PH marks."""
import ctypes, os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
from var_amd._lib import load_library
which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
HW = int(os.environ.get("PH_HW", "84"))      # PH_HW=96: the 96 x 96 forms
cfg = types.SimpleNamespace(img_dim=(3, HW, HW), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m)
pool = var_amd.SyntheticTripletPool(1024, hw=HW, seed=0, clips_per_class=8).freeze_pairs()
tr.ctx.set_streams(0)
def step():
    i, c, l = pool.next_batch_indices(B)
    tr.step_from_dataset(pool.images, i, pool.clips, c, l)
for _ in range(3): step()
torch.cuda.synchronize()
lib = load_library()
fn = getattr(lib, "var_debug_phases_" + which)
buf = (ctypes.c_ulonglong * 32)()
fn(buf)
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
fn(buf)
v = [x / n for x in buf]
tot = sum(v[:16])
print(which, "B", B, "total cycles/launch %.0f (%.1f us at 2.4 GHz)" % (tot, tot / 2400))
if v[16]:
    print("  in-kernel clock: %.0f shader cycles in %.2f us of s_memrealtime = %.3f GHz" % (v[17], v[16] / 100.0, v[17] / v[16] * 0.1))
for i, x in enumerate(v[:16]):
    if x: print("  phase %2d: %8.0f cycles  %5.1f %%" % (i, x, 100 * x / tot))
