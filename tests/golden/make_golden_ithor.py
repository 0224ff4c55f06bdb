#!/usr/bin/env python3
"""Golden vectors for the iTHOR VARPretextNet, made by IMPORTING the
reference class models/pretext/ai2thor_pretext_model.py:VARPretextNet on CPU.

Its constructor calls `.cuda()` on one constant (`zero_feat`, ai2thor_pretext_model.py:46, never used in
forward); this harness binds torch.Tensor.cuda to the identity for the duration of the import so that the class
constructs on a CPU-only host -- the reference itself is untouched.

The model has 3 849 126 parameters (15.4 MB), too many to commit: the fixture stores the SEED (977 =
pretextEnvSeed of Envs/ai2thor/config.py:63) together with per-tensor check values (sum, sum of magnitudes, the
first 8 numbers), and the oracle (oracle/torch_oracle.py:IthorNetCPU) re-creates the weights from the seed with
the same constructor order; tests/test_oracle_ithor.py checks them against these values.  Gradients are stored
as per-tensor L2 norms plus every 101st element.

ithor_h96.npz   u8 images (2,3,96,96), f32 sounds (2,1,600,40) x {pos,neg}, the 7-key forward dict,
                TripletMarginLoss(margin=1, p=2), gradient norms/samples, loss + parameter samples after
                two Adam steps (lr 1e-4, weight_decay 1e-6, VAR/pretext_VAR.py:33-35)

Usage:  python tests/golden/make_golden_ithor.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

SEED = 977
STRIDE = 101


def cfg(h):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 600, 40), representationDim=3)


def make_inputs(b, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(b, 3, h, h), dtype=np.uint8)
    # python_speech_features-like value range: coefficient 0 is a log energy (~ 15..25), the rest O(10)
    snd = rng.standard_normal((2 * b, 1, 600, 40)).astype(np.float32) * 6.0
    snd[:, :, :, 0] += 18.0
    snd[1, :, 400:] = 0.0                                   # zero padding in the MFCC domain (audioLoader.py:249-250)
    return img, snd[:b].copy(), snd[b:].copy()


def check_values(v):
    f = v.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]])


def main():
    torch.set_num_threads(4)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # harness-side shim, see the module docstring
    try:
        from models.pretext.ai2thor_pretext_model import VARPretextNet
        torch.manual_seed(SEED)
        m = VARPretextNet(cfg(96))
    finally:
        torch.Tensor.cuda = real_cuda
    m.train()
    out = {"seed": np.int64(SEED), "stride": np.int64(STRIDE)}
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        out["shape." + k] = np.asarray(v.shape, dtype=np.int64)
        out["check." + k] = check_values(v.numpy())
    out["names"] = np.asarray(names)

    img, pos, neg = make_inputs(2, 96, 11)
    out.update(image=img, sound_positive=pos, sound_negative=neg)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-4, weight_decay=1e-6)
    crit = torch.nn.TripletMarginLoss(margin=1.0, p=2)
    losses = []
    for step in range(2):
        opt.zero_grad()
        image = (torch.from_numpy(img) / 255.).float()
        d = m(image, torch.from_numpy(pos), torch.from_numpy(neg))
        loss = crit(d['image_feat'], d['sound_feat_positive'], d['sound_feat_negative'])
        loss.backward()
        losses.append(loss.item())
        if step == 0:
            for k in ('image_feat', 'sound_feat_positive', 'sound_feat_negative', 'image_feat_raw', 'pos_sound_raw'):
                out[k] = d[k].detach().numpy().copy()
            assert d['image_BCE'] is None and d['sound_BCE'] is None
            for k, p in m.named_parameters():
                g = p.grad.detach().numpy().reshape(-1)
                out["gnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                out["gsamp." + k] = g[::STRIDE].copy()
        opt.step()
    for k, v in m.state_dict().items():
        out["adam2." + k] = v.numpy().reshape(-1)[::STRIDE].copy()
    out["losses"] = np.asarray(losses, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "ithor_h96.npz"), **out)
    print("losses", losses, "keys", len(names))
    print("file bytes", os.path.getsize(os.path.join(HERE, "ithor_h96.npz")))


if __name__ == "__main__":
    main()
