"""Phase timing of the fused actor-critic chain (VAR_HIP_LIB=.../libvar_ph.so): cycles of thread 0 of one workgroup."""
import ctypes, os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
from var_amd._lib import load_library
class Box:
    def __init__(self, n): self.shape = (n,)
B = 8
acfg = types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2)
ac = var_amd.ArmNetPolicy(None, Box(2), config=acfg, base='arm_VAR', base_kwargs={'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512, 'actionHiddenSize': 128}).to("cuda")
obs = {'image': torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device="cuda"), 'image_feat': torch.randn(B, 3, device="cuda"), 'robot_pose': torch.randn(B, 2, device="cuda"), 'goal_sound_feat': torch.randn(B, 3, device="cuda")}
hxs, masks = torch.zeros(B, 512, device="cuda"), torch.ones(B, 1, device="cuda")
for _ in range(3): ac._base_forward(obs, hxs, masks)
torch.cuda.synchronize()
lib = load_library()
buf = (ctypes.c_ulonglong * 32)()
lib.var_debug_phases_armchain(buf)
n = 10
for _ in range(n): ac._base_forward(obs, hxs, masks)
torch.cuda.synchronize()
lib.var_debug_phases_armchain(buf)
v = [x / n for x in buf]
tot = sum(v)
print("chain total cycles/launch %.0f" % tot)
for i, x in enumerate(v):
    if x: print("  phase %2d: %8.0f cycles  %5.1f %%" % (i, x, 100 * x / tot))
