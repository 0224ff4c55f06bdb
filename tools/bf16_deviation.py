import sys, types, numpy as np, torch
sys.path.insert(0, '/root/repo')
import var_amd
fx = dict(np.load('/root/repo/tests/golden/ithor_h96.npz'))
cuda = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
cfg = types.SimpleNamespace(img_dim=(3, 96, 96), sound_dim=(1, 600, 40), representationDim=3)
res = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(977)
    m = var_amd.IthorVARPretextNet(cfg).to("cuda").set_precision(prec)
    tr = var_amd.IthorTrainer(m)
    loss, feats = tr.loss_and_grads(cuda(fx["image"]), cuda(fx["sound_positive"]), cuda(fx["sound_negative"]), feats=True)
    res[prec] = (loss.item(), feats.cpu().numpy().copy(), tr.grads.cpu().numpy().copy())
print("loss", res["fp32"][0], res["bf16"][0], "ref", fx["losses"][0])
print("feat max abs dev", np.abs(res["fp32"][1] - res["bf16"][1]).max())
g0, g1 = res["fp32"][2], res["bf16"][2]
print("grad l2 rel", np.linalg.norm(g0 - g1) / np.linalg.norm(g0))
o = 0
torch.manual_seed(977)
m = var_amd.IthorVARPretextNet(cfg)
worst = 0
for k, p in m.named_parameters():
    a, b = g0[o:o + p.numel()], g1[o:o + p.numel()]
    o += p.numel()
    e = np.linalg.norm(a - b) / (np.linalg.norm(a) + 1e-30)
    worst = max(worst, e)
    if e > 0.03: print(k, e)
print("worst tensor", worst)
