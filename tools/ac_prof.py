import sys, types, torch
sys.path.insert(0, '/root/repo')
import var_amd
class Box:
    def __init__(self, n): self.shape = (n,)
B = 8
acfg = types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2)
ac = var_amd.ArmNetPolicy(None, Box(2), config=acfg, base='arm_VAR', base_kwargs={'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512, 'actionHiddenSize': 128}).to("cuda")
obs = {'image': torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device="cuda"), 'image_feat': torch.randn(B, 3, device="cuda"), 'robot_pose': torch.randn(B, 2, device="cuda"), 'goal_sound_feat': torch.randn(B, 3, device="cuda")}
hxs, masks = torch.zeros(B, 512, device="cuda"), torch.ones(B, 1, device="cuda")
for _ in range(10):
    ac._base_forward(obs, hxs, masks)
torch.cuda.synchronize()
