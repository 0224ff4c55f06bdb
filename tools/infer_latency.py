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
