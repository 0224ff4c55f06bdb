import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import var_amd
from oracle import mfcc_np
from oracle.torch_oracle import KukaNetCPU
from oracle import var_oracle as orc
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sd = dict(np.load(os.path.join(G, "kuka_weights.npz")))
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
B, steps = 256, 4
pool = var_amd.SyntheticTripletPool(768, hw=84, seed=21, clips_per_class=3).freeze_pairs()
table = pool.index_table(B, steps, drop_last=True)[:steps].contiguous()
def new():
    m = var_amd.VARPretextNet(cfg); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to("cuda")
    return m, var_amd.VARTrainer(m, lr=1e-4, weight_decay=1e-6)
def summarize(tag, g, gref):
    u, r = orc.unflatten_params(g), orc.unflatten_params(gref)
    print(tag, {k.split('.')[0][:4] + k.split('.')[1] + k.split('.')[2][0]: "%.1e/%.1e" % (np.max(np.abs(u[k] - r[k])), np.max(np.abs(r[k]))) for k in list(u)[:10:2] + list(u)[18:20]})
# A: eager pcm path, B: step() with HIP-computed features, C: torch reference at the same (A's) weights each step
ma, ta = new(); mb, tb = new()
for s in range(steps):
    r = table[s]
    f = var_amd.mfcc(pool.clips, r[3*B:], out_frames=100, clip_index=r[B:3*B])
    img = pool.images[r[:B].long()].contiguous()
    # torch reference gradient at A's current weights with HIP features
    net = KukaNetCPU(); net.load_state_dict({k: v.detach().cpu().clone() for k, v in ma.state_dict().items()})
    a, p, n = net((img.cpu() / 255.).float(), f[:B].cpu(), f[B:].cpu())
    torch.nn.TripletMarginLoss()(a, p, n).backward()
    gref = np.concatenate([dict(net.named_parameters())[k].grad.reshape(-1).numpy() for k, _ in var_amd.PARAM_SPECS])
    ta.step_from_dataset(pool.images, r[:B], pool.clips, r[B:3*B], r[3*B:])
    ga = ta.grads.cpu().numpy().copy()
    tb.step(img, f[:B].contiguous(), f[B:].contiguous())
    gb = tb.grads.cpu().numpy().copy()
    print("step", s, "A==B grads", np.array_equal(ga, gb), "params equal", torch.equal(ma.flat_parameters(), mb.flat_parameters()))
    summarize("  A vs torch (maxerr/maxref)", ga, gref)
    summarize("  B vs torch", gb, gref)
