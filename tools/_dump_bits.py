import os, sys, types
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import var_amd
from var_amd._lib import Context
B = 19
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
sd = dict(np.load("/root/repo/tests/golden/kuka_weights2.npz"))
model = var_amd.VARPretextNet(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to("cuda")
tr = var_amd.VARTrainer(model)
rng = np.random.default_rng(5)
img = torch.from_numpy(rng.integers(0, 256, size=(B, 3, 84, 84), dtype=np.uint8)).cuda()
g0 = torch.Generator().manual_seed(1)
pos = torch.randn(B, 1, 100, 40, generator=g0).cuda(); neg = torch.randn(B, 1, 100, 40, generator=g0).cuda()
tr.loss_and_grads(img, pos, neg)
torch.cuda.synchronize()
ctx = Context.get(0)
out = {}
for l, ch, h in ((1, 32, 42), (2, 32, 21), (3, 64, 11)):
    out[f"act{l}"] = ctx.debug_buffer(f"act{l}").cpu().numpy()[:B * ch * h * h].reshape(B, ch, h, h)
    out[f"gact{l}"] = ctx.debug_buffer(f"gact{l}").cpu().numpy()[:B * ch * h * h].reshape(B, ch, h, h)
out["bits"] = ctx.debug_buffer("relu1").cpu().numpy().view(np.uint16)[:B * 2 * 42 * 42].reshape(B, 2, 42, 42)
out["g"] = tr.grads.cpu().numpy()
np.savez(sys.argv[1], **out)
