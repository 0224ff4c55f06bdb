"""Phase timing of the bf16 iTHOR kernels in an instrumented build (make phases; VAR_HIP_LIB=.../libvar_ph.so):
cycles of thread 0 of workgroup 0 between PH marks, per training step."""
import ctypes, os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
from var_amd._lib import load_library
which = sys.argv[1] if len(sys.argv) > 1 else "snd"
B = 256
cfg = types.SimpleNamespace(img_dim=(3, 96, 96), sound_dim=(1, 600, 40), representationDim=3)
torch.manual_seed(977)
m = var_amd.IthorVARPretextNet(cfg).to("cuda").set_precision("bf16")
tr = var_amd.IthorTrainer(m)
g = torch.Generator(device="cuda").manual_seed(0)
img = torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device="cuda", generator=g)
pos = torch.randn(B, 1, 600, 40, device="cuda", generator=g) * 3
neg = torch.randn(B, 1, 600, 40, device="cuda", generator=g) * 3
for _ in range(2): tr.step(img, pos, neg)
torch.cuda.synchronize()
lib = load_library()
fn = getattr(lib, "var_debug_phases_" + which)
buf = (ctypes.c_ulonglong * 32)()
fn(buf)
n = 5
for _ in range(n): tr.step(img, pos, neg)
torch.cuda.synchronize()
fn(buf)
v = [x / n for x in buf]
tot = sum(v)
print(which, "total cycles/step %.0f" % tot)
for i, x in enumerate(v):
    if x: print("  phase %2d: %10.0f  %5.1f %%" % (i, x, 100 * x / tot))
