"""Exploration for tests/test_gpu_round3.py::test_ithor_bf16_gradient_drift_is_traced_to_routing_flips: split the bf16 - fp32
difference of the image branch's bias gradients into the part that sits on units the gradient reaches in only one of the two runs
(ReLU gate / pool winner changed hands) and the part on units both runs reach (arithmetic)."""
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import Context
from oracle.torch_oracle import IthorNetCPU
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_round3 import _ithor_batch, icfg, _param_spans
for seed in (43, 44, 45):
    B = 8
    img, pos, neg = (t.cuda() for t in _ithor_batch(B, seed))
    torch.manual_seed(977)
    sd = IthorNetCPU().state_dict()
    ga, grads = {}, {}
    side = {1: 96, 2: 96, 3: 48, 4: 24, 5: 12}              # conv l's output map (hs[0], hs[0], hs[1], hs[2], hs[3])
    ch = [3, 32, 32, 64, 64, 128, 128]
    for prec in ("fp32", "bf16"):
        m = var_amd.IthorVARPretextNet(icfg(96)); m.load_state_dict(sd)
        m = m.to("cuda").set_precision(prec, keep_fp32_activations=True)
        tr = var_amd.IthorTrainer(m); tr.loss_and_grads(img, pos, neg); torch.cuda.synchronize()
        ctx = Context.get(0)
        ga[prec] = {}
        for l in range(1, 6):
            hw = side[l]
            n = B * ch[l] * hw * hw
            ga[prec][l] = ctx.debug_buffer(f"ithor_ga{l}")[:n].clone().view(B, ch[l], hw * hw)
        grads[prec] = tr.grads.clone(); spans = _param_spans(m)
    for l in range(1, 6):
        k = f"imgBranch.{[0, 2, 5, 8, 11][l - 1]}.bias"
        lo, hi = spans[k]
        b32, b16 = grads["fp32"][lo:hi], grads["bf16"][lo:hi]
        g32, g16 = ga["fp32"][l], ga["bf16"][l]
        s32, s16 = g32.sum((0, 2)), g16.sum((0, 2))
        diff_route = (g32 != 0) != (g16 != 0)
        d = g16 - g32
        d_route = torch.where(diff_route, d, torch.zeros_like(d)).sum((0, 2))
        d_arith = torch.where(diff_route, torch.zeros_like(d), d).sum((0, 2))
        nb = float(b32.norm())
        print(f"seed {seed} {k}: drift {float((b16-b32).norm())/nb:.3f} | chan_sum check {float((s32-b32).norm())/nb:.1e} {float((s16-b16).norm())/nb:.1e}"
              f" | routed-differently units {int(diff_route.sum())} ({float(diff_route.float().mean()):.2e}) | route part {float(d_route.norm())/nb:.3f}"
              f" arith part {float(d_arith.norm())/nb:.3f} | cancellation sum|g|/|sum g| {float(g32.abs().sum((0,2)).norm())/nb:.1f}")
