#!/usr/bin/env python3
"""Per-kernel-family timings of one training step (B=256, 84x84 or --hw 96) with every launch on ONE stream
(var_set_streams(0)), measured with the library's own HIP-event hooks (var_profile_select/read).
Usage: python tools/kbench.py [--batch 256] [--steps 20] [--tags 0,1,2] [--hw 96]"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import var_amd  # noqa: E402
from var_amd._lib import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--tags", type=str, default="")
ap.add_argument("--hw", type=int, default=84)
args = ap.parse_args()
B = args.batch
cfg = types.SimpleNamespace(img_dim=(3, args.hw, args.hw), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
model = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(model)
pool = var_amd.SyntheticTripletPool(int(os.environ.get("KB_POOL", "2048")), hw=args.hw, seed=0, clips_per_class=int(os.environ.get("KB_CPC", "32"))).freeze_pairs()
ctx = Context.get(0)
ctx.set_streams(0)


def step():
    i, c, l = pool.next_batch_indices(B)
    tr.step_from_dataset(pool.images, i, pool.clips, c, l)


for _ in range(5):
    step()
names = ctx.tag_names()
tags = [int(t) for t in args.tags.split(",")] if args.tags else range(len(names))
total = 0.0
for t in tags:
    ctx.profile_select(t)
    for _ in range(args.steps):
        step()
    ms, n = ctx.profile_read()
    if n:
        per_step = 1e3 * ms / args.steps
        total += per_step
        print(f"{names[t]:28s} {1e3 * ms / n:9.1f} us/launch  {per_step:9.1f} us/step  ({n // args.steps} launches/step)")
ctx.profile_select(-1)
print(f"{'sum':28s} {'':9s}            {total:9.1f} us/step")
