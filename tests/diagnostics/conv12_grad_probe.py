"""Which image-CNN gradient tensors differ from torch at 96 x 96 (debug of the tail kernel)."""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import var_amd
from tests._gpu_helpers import torch_loss_grad
from oracle import var_oracle as orc      # (debug probe: the checker's parameter order)
h = int(sys.argv[1]) if len(sys.argv) > 1 else 96
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 100, 40), representationDim=3)
torch.manual_seed(1)
m = var_amd.VARPretextNet(cfg).to("cuda")
tr = var_amd.VARTrainer(m)
rng = np.random.default_rng(7)
img = rng.integers(0, 256, size=(B, 3, h, h), dtype=np.uint8)
pos = rng.standard_normal((B, 1, 100, 40)).astype(np.float32)
neg = rng.standard_normal((B, 1, 100, 40)).astype(np.float32)
sd0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
FLT = len(sys.argv) > 3
img_dev = (torch.from_numpy(img) / 255.).float().cuda() if FLT else torch.from_numpy(img).cuda()
tr.loss_and_grads(img_dev, torch.from_numpy(pos).cuda(), torch.from_numpy(neg).cuda())
net, l_ref, g_ref, _, image_f32 = torch_loss_grad(sd0, torch.from_numpy(img), torch.from_numpy(pos), torch.from_numpy(neg), h)
g = tr.grads.cpu().numpy()
off = 0
for k, shp in orc.PARAM_SPECS:
    n = int(np.prod(shp))
    a, b = g[off:off + n], g_ref[off:off + n]
    off += n
    if k.startswith("imgBranch.0") or k.startswith("imgBranch.2"):
        print(k, "rel err %.3e" % (np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)))
        if k == "imgBranch.2.weight":
            d = (a - b).reshape(32, 32, 3, 3); r = b.reshape(32, 32, 3, 3)
            print("  per tap:", np.round(np.sqrt((d ** 2).sum((0, 1)) / (r ** 2).sum((0, 1))), 3).tolist())
            print("  per out-channel n:", np.round(np.sqrt((d ** 2).sum((1, 2, 3)) / (r ** 2).sum((1, 2, 3))), 2).tolist())
            print("  got[n=0,c=14..18,1,1]", a.reshape(32, 32, 3, 3)[0, 14:19, 1, 1].tolist(), "ref", r[0, 14:19, 1, 1].tolist())
            print("  got[n=17,c=14..18,0,0]", a.reshape(32, 32, 3, 3)[17, 14:19, 0, 0].tolist(), "ref", r[17, 14:19, 0, 0].tolist())
            print("  per in-channel c:", np.round(np.sqrt((d ** 2).sum((0, 2, 3)) / (r ** 2).sum((0, 2, 3))), 2).tolist())
        if k == "imgBranch.0.weight":
            d = (a - b).reshape(32, 3, 3, 3)
            r = b.reshape(32, 3, 3, 3)
            print("  per tap (ky,kx) err/ref:", np.round(np.linalg.norm(d, axis=(0, 1)) / np.linalg.norm(r, axis=(0, 1)), 3).tolist())
            print("  per out-channel:", np.round(np.sqrt((d ** 2).sum((1, 2, 3)) / (r ** 2).sum((1, 2, 3))), 2).tolist())
            print("  per in-channel:", np.round(np.sqrt((d ** 2).sum((0, 2, 3)) / (r ** 2).sum((0, 2, 3))), 2).tolist())
