#!/usr/bin/env python3
"""Throughput of the iTHOR pretext step (BASELINE config 4 shape: img 96x96, sound (1,600,40)) on one GPU:
var_ithor_loss_grad + var_adam_step per step, synthetic inputs resident in HBM, fp32.
  python tools/ithor_bench.py [--batch 256] [--steps 10] [--warmup 3] [--hw 96]"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd  # noqa: E402

# 2*MAC per triplet, fwd + bwd (fwd 3318 MFLOP; no dX for the two first convolutions)
FLOP_FWD = 3318e6
FLOP_STEP = 3 * FLOP_FWD - 2 * 96 * 96 * 27 * 32 - 2 * (2 * 300 * 20 * 121 * 64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hw", type=int, default=96)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32", help="operand precision of the products")
    ap.add_argument("--features", action="store_true", help="feed precomputed (1,600,40) features instead of PCM")
    a = ap.parse_args()
    torch.manual_seed(977)
    cfg = types.SimpleNamespace(img_dim=(3, a.hw, a.hw), sound_dim=(1, 600, 40), representationDim=3)
    m = var_amd.IthorVARPretextNet(cfg).to("cuda")
    m.set_precision("bf16" if a.dtype == "bf16" else "fp32")
    tr = var_amd.IthorTrainer(m)
    g = torch.Generator(device="cuda").manual_seed(0)
    img = torch.randint(0, 256, (a.batch, 3, a.hw, a.hw), dtype=torch.uint8, device="cuda", generator=g)
    pos = torch.randn((a.batch, 1, 600, 40), device="cuda", generator=g) * 6
    neg = torch.randn((a.batch, 1, 600, 40), device="cuda", generator=g) * 6
    # 6 s int16 clips resident in HBM (Envs/ai2thor: FSC utterances up to 96000 samples), ragged lengths
    pcm = torch.randint(-8000, 8000, (2 * a.batch, 96000), dtype=torch.int16, device="cuda", generator=g)
    lens = torch.randint(30000, 96001, (2 * a.batch,), dtype=torch.int32, device="cuda", generator=g)
    lens[::5] = 0                                            # 20 % "empty" class

    def one():
        if a.fwd_only:
            with torch.no_grad():
                m(img, pos, neg)
        elif a.features:
            tr.step(img, pos, neg)
        else:
            tr.step_from_pcm(img, pcm, lens)                 # MFCC front-end inside the step

    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        one()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    flop = FLOP_FWD if a.fwd_only else FLOP_STEP
    print(json.dumps({"workload": "ithor pretext step" if not a.fwd_only else "ithor forward", "batch": a.batch,
                      "hw": a.hw, "ms_per_step": round(ms, 3), "triplets_per_s": round(a.batch / ms * 1e3, 1),
                      "tflops": round(a.batch * flop / ms / 1e9, 2), "dtype": a.dtype,
                      "loss": float(tr.loss.item()) if not a.fwd_only else None}))


if __name__ == "__main__":
    main()
