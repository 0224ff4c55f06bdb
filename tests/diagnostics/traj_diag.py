"""Diagnostic: 10-step trajectory HIP (replayed graph) vs CPU restatement, per-tensor drift, run twice."""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from oracle import mfcc_np
from oracle.torch_oracle import CPUTrainer
from oracle import var_oracle as orc
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
sd = dict(np.load(os.path.join(G, "kuka_weights.npz")))
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
B, steps = 256, 10
pool = var_amd.SyntheticTripletPool(768, hw=84, seed=21, clips_per_class=3).freeze_pairs()
table = pool.index_table(B, steps, drop_last=True)[:steps].contiguous()
def feats(idx, lens):
    out = np.zeros((len(idx), 1, 100, 40), np.float32)
    clips = pool.clips.cpu().numpy()
    cache = {}
    for i, (c, n) in enumerate(zip(idx.tolist(), lens.tolist())):
        if n > 0:
            if (c, n) not in cache:
                cache[(c, n)] = mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(clips[c, :n]).astype(np.float32))
            out[i] = cache[(c, n)]
    return torch.from_numpy(out)
def hip_run(mode):
    m = var_amd.VARPretextNet(cfg); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to("cuda")
    tr = var_amd.VARTrainer(m, lr=1e-4, weight_decay=1e-6)
    losses = []
    if mode == "graph":
        replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
        for s in range(steps): losses.append(float(replay().item()))
    elif mode == "eager":
        for s in range(steps):
            r = table[s]
            losses.append(float(tr.step_from_dataset(pool.images, r[:B], pool.clips, r[B:3*B], r[3*B:]).item()))
    else:   # eager with oracle features
        tc = table.cpu()
        for s in range(steps):
            r = tc[s]
            losses.append(float(tr.step(pool.images[r[:B].long().cuda()].contiguous(), feats(r[B:2*B], r[3*B:4*B]).cuda(), feats(r[2*B:3*B], r[4*B:]).cuda()).item()))
    return losses, m.flat_parameters().cpu().numpy().copy()
ref = CPUTrainer(state_dict=sd, lr=1e-4, weight_decay=1e-6)
tc = table.cpu(); lref = []
for s in range(steps):
    r = tc[s]
    lref.append(ref.step(pool.images[r[:B].long().cuda()].cpu(), feats(r[B:2*B], r[3*B:4*B]), feats(r[2*B:3*B], r[4*B:])))
pref = np.concatenate([ref.model.state_dict()[k].reshape(-1).numpy() for k, _ in var_amd.PARAM_SPECS])
runs = {}
for mode in ("graph", "graph", "eager", "oraclefeat"):
    l, p = hip_run(mode)
    print(mode, "max loss diff", max(abs(a - b) for a, b in zip(l, lref)))
    d = np.abs(p - pref)
    print("   frac<2e-6 %.4f max %.3e" % (np.mean(d < 2e-6), d.max()))
    du = orc.unflatten_params(d)
    print("   per tensor frac>2e-6:", {k.replace('weight','w').replace('bias','b'): round(float(np.mean(v > 2e-6)), 3) for k, v in du.items()})
    if mode in runs:
        print("   same-mode rerun identical:", np.array_equal(runs[mode], p))
    runs[mode] = p
print("graph vs eager identical:", np.array_equal(runs["graph"], runs["eager"]), np.abs(runs["graph"] - runs["eager"]).max())
