#!/usr/bin/env python3
"""BASELINE config 5: latency of the frozen-encoder step of the RL stage (8 envs): embeddings of 8 images + 8 goal
sounds and the intrinsic reward, through the HIP forward.  Prints us per call (eager launches; steady state)."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import var_amd  # noqa: E402

cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda").eval()
B = 8
img = torch.randint(0, 256, (B, 3, 84, 84), dtype=torch.uint8, device="cuda")
goal = torch.randn(B, 1, 100, 40, device="cuda")
inf = torch.full((B, 1, 100, 40), float("inf"), device="cuda")
with torch.no_grad():
    for name, snd in (("image + goal sound", goal), ("image only (goal sound cached)", inf)):
        for _ in range(20):
            d = m(img, snd, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            d = m(img, snd, None)
            r = (d["image_feat"] * d["sound_feat_positive"]).sum(1)
        torch.cuda.synchronize()
        print(f"{name:34s} {1e6 * (time.perf_counter() - t0) / n:8.1f} us per step (B={B}, eager, device-side reward)")

r = var_amd.IntrinsicReward(m).capture(B)
for name, snd in (("graph: image + goal sound", goal), ("graph: image only (goal cached)", None)):
    for _ in range(20):
        r.step(img, snd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 1000
    for _ in range(n):
        r.step(img, snd)
    torch.cuda.synchronize()
    print(f"{name:34s} {1e6 * (time.perf_counter() - t0) / n:8.1f} us per step (B={B}, replayed graph incl. input copies)")

# ---- second half of config 5: the actor-critic forward (Policy.act, models/ppo/model.py:57-69) ----
class _Box:
    def __init__(self, n):
        self.shape = (n,)


_Box.__name__ = "Box"
acfg = types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2)
torch.manual_seed(453)
ac = var_amd.ArmNetPolicy(None, _Box(2), config=acfg, base='arm_VAR',
                          base_kwargs={'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512,
                                       'actionHiddenSize': 128}).to("cuda")
obs = {'image': torch.randint(0, 256, (B, 3, 96, 96), dtype=torch.uint8, device="cuda"),
       'image_feat': torch.randn(B, 3, device="cuda"), 'robot_pose': torch.randn(B, 2, device="cuda"),
       'goal_sound_feat': torch.randn(B, 3, device="cuda")}
hxs, masks = torch.zeros(B, 512, device="cuda"), torch.ones(B, 1, device="cuda")
for _ in range(20):
    v, a, lp, hxs = ac.act(obs, hxs, masks)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    v, a, lp, hxs = ac.act(obs, hxs, masks)
torch.cuda.synchronize()
print(f"{'actor-critic act()':34s} {1e6 * (time.perf_counter() - t0) / n:8.1f} us per step (B={B}, eager, incl. sampling)")
# the same forward as a replayed graph over static buffers
g = var_amd._lib.new_graph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ac._base_forward(obs, hxs, masks)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        outs = ac._base_forward(obs, hxs, masks)
torch.cuda.synchronize()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 1000
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print(f"{'graph: actor-critic forward':34s} {1e6 * (time.perf_counter() - t0) / n:8.1f} us per step (B={B}, replayed graph)")
