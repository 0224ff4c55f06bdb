"""Frozen-encoder reward step (BASELINE config 5, first half) as a replayed graph, for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace -d gpurun_out/prof_rw -o rw -f csv -- python3 tools/reward_trace.py goal|cached"""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda").eval()
B = 8
img = torch.randint(0, 256, (B, 3, 84, 84), dtype=torch.uint8, device="cuda")
goal = torch.randn(B, 1, 100, 40, device="cuda")
r = var_amd.IntrinsicReward(m).capture(B)
mode = sys.argv[1]
for _ in range(60):
    r.step(img, goal if mode == "goal" else None)
torch.cuda.synchronize()
