import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import var_amd
from oracle import mfcc_np
from oracle.torch_oracle import CPUTrainer
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
sd = dict(np.load(os.path.join(G, "kuka_weights.npz")))
cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
B, steps = 256, 10
pool = var_amd.SyntheticTripletPool(768, hw=84, seed=21, clips_per_class=3).freeze_pairs()
table = pool.index_table(B, steps, drop_last=True)[:steps].contiguous()
tc = table.cpu()
clips = pool.clips.cpu().numpy()
cache = {}
def ofeats(idx, lens):
    out = np.zeros((len(idx), 1, 100, 40), np.float32)
    for i, (c, n) in enumerate(zip(idx.tolist(), lens.tolist())):
        if n > 0:
            if (c, n) not in cache:
                cache[(c, n)] = mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(clips[c, :n]).astype(np.float32))
            out[i] = cache[(c, n)]
    return torch.from_numpy(out)
F_or = [(ofeats(tc[s][B:2*B], tc[s][3*B:4*B]), ofeats(tc[s][2*B:3*B], tc[s][4*B:])) for s in range(steps)]
F_hip = []
for s in range(steps):
    f = var_amd.mfcc(pool.clips, table[s][3*B:], out_frames=100, clip_index=table[s][B:3*B]).cpu()
    F_hip.append((f[:B].contiguous(), f[B:].contiguous()))
print("feature diff max", max(float((a[0] - b[0]).abs().max()) for a, b in zip(F_or, F_hip)))
imgs = [pool.images[tc[s][:B].long().cuda()].contiguous() for s in range(steps)]
def hip(F):
    m = var_amd.VARPretextNet(cfg); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.to("cuda")
    tr = var_amd.VARTrainer(m, lr=1e-4, weight_decay=1e-6)
    for s in range(steps): tr.step(imgs[s], F[s][0].cuda(), F[s][1].cuda())
    return m.flat_parameters().cpu().numpy().copy()
def cpu(F, threads):
    torch.set_num_threads(threads)
    ref = CPUTrainer(state_dict=sd, lr=1e-4, weight_decay=1e-6)
    for s in range(steps): ref.step(imgs[s].cpu(), F[s][0], F[s][1])
    return np.concatenate([ref.model.state_dict()[k].reshape(-1).numpy() for k, _ in var_amd.PARAM_SPECS])
R = {"hip/or": hip(F_or), "hip/hip": hip(F_hip), "cpu/or": cpu(F_or, 16), "cpu/hip": cpu(F_hip, 16), "cpu1/or": cpu(F_or, 1), "cpu4/or": cpu(F_or, 4)}
ks = list(R)
for i in range(len(ks)):
    for j in range(i + 1, len(ks)):
        d = np.abs(R[ks[i]] - R[ks[j]])
        print("%-8s vs %-8s frac<2e-6 %.4f  max %.2e" % (ks[i], ks[j], np.mean(d < 2e-6), d.max()))
