"""Debug: repeat the B=256 loss+grad step and compare act3/act4/act5/hid_i between runs and against the forward-only call."""
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from var_amd._lib import Context
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(453)
m = var_amd.VARPretextNet(cfg).to("cuda")
pool = var_amd.SyntheticTripletPool(512, hw=84, seed=3, clips_per_class=8)
idx, cp = pool.sample_indices(256)
img, pcm, lens = pool.gather(idx, cp)
feats = var_amd.mfcc(pcm, lens)
pos, neg = feats[:256].contiguous(), feats[256:].contiguous()
ctx = Context.get(0)
with torch.no_grad():
    full = m(img, pos, neg)
torch.cuda.synchronize()
ref = {k: ctx.debug_buffer(k).clone() for k in ("act2", "act3", "act4", "act5", "hid_i")}
tr = var_amd.VARTrainer(m)
for it in range(6):
    tr.loss_and_grads(img, pos, neg)
    torch.cuda.synchronize()
    msg = [f"loss {tr.loss.item():.7f}"]
    for k in ref:
        d = (ctx.debug_buffer(k) - ref[k]).abs()
        nb = int((d.view(256, -1).amax(1) > 0).sum())
        msg.append(f"{k}: max {d.max().item():.3e} imgs {nb}")
        if nb and k in ("act3", "act4", "act5", "hid_i"):
            bad = torch.nonzero(d.view(256, -1).amax(1) > 0).flatten()[:6].tolist()
            e = torch.nonzero(d.view(256, -1)[bad[0]] > 0).flatten()
            msg.append(f"   first bad imgs {bad}; in img {bad[0]}: {e.numel()} elems, first {e[:8].tolist()}")
    print(" | ".join(msg))
