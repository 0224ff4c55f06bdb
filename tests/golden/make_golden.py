#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference model classes on CPU.

Runs only in the build container (needs /root/reference); the GPU box sees only
the .npz files this writes.  Nothing from the reference is copied: the files
hold inputs and the numbers the reference computed from them.

What is recorded (SURVEY.md section 8c "How the oracle is taken"):
  kuka_weights.npz      state_dict of models.pretext.arm_pretext_model.VARPretextNet,
                        torch.manual_seed(453) (pretextEnvSeed, fourInARow/config.py:55)
  kuka_h{84,96}.npz     u8 images, f32 MFCC pos/neg (with "empty" all-zero rows),
                        the 7-key forward dict, TripletMarginLoss(margin=1,p=2),
                        every parameter .grad
  kuka_weights2.npz / kuka_h84_w2.npz   the same for a scaled ("spread-out") weight set, so
                        that some hinge terms are inactive and gradients are O(1)
  kuka_adam.npz         losses and parameters after 1 and 3 Adam steps
                        (lr 1e-4, weight_decay 1e-6: VAR/pretext_VAR.py:33-35)
  kuka_edge.npz         None-input / cached-sound (all-inf) behaviours of
                        models/pretext/pretext_base.py:10-41
  lr_schedule.npz       MultiStepLR([10,30,50], 0.2) sequence (utils.py:42-46)
  reward_norm.npz       a stream of per-env rewards / episode ends pushed through the reference's RunningMeanStd
                        (Envs/vec_env/running_mean_std.py, loaded by file path) with the six lines of
                        VecPretextNormalize.step_wait that normalise the reward (vec_pretext_normalize.py:52-59)

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from models.pretext.arm_pretext_model import VARPretextNet  # noqa: E402  (reference)
from utils import get_scheduler  # noqa: E402  (reference utils.py)
from oracle import mfcc_np  # noqa: E402  (only to make realistic MFCC inputs)


def cfg(h):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 100, 40), representationDim=3)


def make_inputs(b, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(b, 3, h, h), dtype=np.uint8)
    clips = mfcc_np.synth_clips(2 * b, seed=seed + 1)
    feats = np.stack([mfcc_np.process_sound_feat(mfcc_np.mfcc_torchaudio(c).astype(np.float32))
                      for c in clips])                      # (2b,1,100,40)
    pos, neg = feats[:b].copy(), feats[b:].copy()
    # the "empty" class is an all-zero MFCC (dataset.py:37-38,58-59)
    pos[1] = 0.0
    neg[b - 1] = 0.0
    return img, pos.astype(np.float32), neg.astype(np.float32)


def np_sd(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def run_fwd_bwd(model, img_u8, pos, neg):
    model.zero_grad()
    image = (torch.from_numpy(img_u8) / 255.).float()       # dataset.py:67-68
    d = model(image, torch.from_numpy(pos).float(), torch.from_numpy(neg).float())
    crit = torch.nn.TripletMarginLoss(margin=1.0, p=2)      # VAR/pretext_VAR.py:38
    loss = crit(d['image_feat'], d['sound_feat_positive'], d['sound_feat_negative'])
    loss.backward()
    return d, loss


def main():
    torch.set_num_threads(1)
    torch.manual_seed(453)
    model = VARPretextNet(cfg(84))
    model.train()
    sd0 = np_sd(model)
    np.savez(os.path.join(HERE, "kuka_weights.npz"), **sd0)

    for h, b, seed in ((84, 6, 100), (96, 5, 200)):
        m = VARPretextNet(cfg(h))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
        m.train()
        img, pos, neg = make_inputs(b, h, seed)
        d, loss = run_fwd_bwd(m, img, pos, neg)
        out = dict(image=img, sound_positive=pos, sound_negative=neg, loss=np.float32(loss.item()))
        for k in ('image_feat', 'sound_feat_positive', 'sound_feat_negative',
                  'image_feat_raw', 'pos_sound_raw'):
            out[k] = d[k].detach().numpy()
        assert d['image_BCE'] is None and d['sound_BCE'] is None
        for k, p in m.named_parameters():
            out['grad.' + k] = p.grad.detach().numpy().copy()
        np.savez(os.path.join(HERE, f"kuka_h{h}.npz"), **out)
        print(f"h={h} loss={loss.item():.7f}")

    # --- a second, "spread-out" weight set: the seeded init scaled so that embeddings
    # differ strongly between samples and some hinge terms are inactive (loss != margin)
    rng = np.random.default_rng(7)
    sd1 = {}
    for k, v in sd0.items():
        if k.endswith('weight'):
            sd1[k] = (v * (3.0 if 'Triplet' in k else 1.6)).astype(np.float32)
        else:
            sd1[k] = (v + 0.05 * rng.standard_normal(v.shape)).astype(np.float32)
    np.savez(os.path.join(HERE, "kuka_weights2.npz"), **sd1)
    m = VARPretextNet(cfg(84))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd1.items()})
    m.train()
    img, pos, neg = make_inputs(7, 84, 500)
    d, loss = run_fwd_bwd(m, img, pos, neg)
    out = dict(image=img, sound_positive=pos, sound_negative=neg, loss=np.float32(loss.item()))
    for k in ('image_feat', 'sound_feat_positive', 'sound_feat_negative',
              'image_feat_raw', 'pos_sound_raw'):
        out[k] = d[k].detach().numpy()
    for k, p in m.named_parameters():
        out['grad.' + k] = p.grad.detach().numpy().copy()
    np.savez(os.path.join(HERE, "kuka_h84_w2.npz"), **out)
    print(f"w2 loss={loss.item():.7f}")
    print(d['image_feat'].detach().numpy(), d['sound_feat_positive'].detach().numpy())

    # --- Adam trajectory (same hyper-parameters as VAR/pretext_VAR.py:33-35) ---
    m = VARPretextNet(cfg(84))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    m.train()
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-4, weight_decay=1e-6)
    out = {}
    losses = []
    batches = [make_inputs(6, 84, 300 + 7 * s) for s in range(3)]
    for s in range(3):
        img, pos, neg = batches[s]
        out[f'image{s}'], out[f'pos{s}'], out[f'neg{s}'] = img, pos, neg
        opt.zero_grad()
        d, loss = run_fwd_bwd(m, img, pos, neg)
        opt.step()
        losses.append(loss.item())
        if s in (0, 2):
            for k, v in np_sd(m).items():
                out[f'step{s + 1}.' + k] = v
    out['losses'] = np.asarray(losses, dtype=np.float32)
    np.savez(os.path.join(HERE, "kuka_adam.npz"), **out)
    print("adam losses", losses)

    # --- edge behaviours of PretextNetBase.VAR_forward ---
    m = VARPretextNet(cfg(84))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    m.eval()
    img, pos, neg = make_inputs(4, 84, 400)
    image = (torch.from_numpy(img) / 255.).float()
    out = dict(image=img, sound_positive=pos, sound_negative=neg)
    with torch.no_grad():
        d = m(image, torch.from_numpy(pos), None)           # pretext.py:131,188
        assert d['sound_feat_negative'] is None
        out['a.image_feat'] = d['image_feat'].numpy()
        out['a.sound_feat_positive'] = d['sound_feat_positive'].numpy()
        out['a.pos_sound_raw'] = d['pos_sound_raw'].numpy()
        # all-inf goal sound => cached embedding is returned, pos_sound_raw None
        inf = torch.full_like(torch.from_numpy(pos), float('inf'))
        d = m(image, inf, None)                             # pretext_base.py:29-32
        assert d['pos_sound_raw'] is None
        out['b.sound_feat_positive'] = d['sound_feat_positive'].numpy()
        out['b.image_feat'] = d['image_feat'].numpy()
        d = m(None, torch.from_numpy(neg), None)            # image=None
        assert d['image_feat'] is None and d['image_feat_raw'] is None
        out['c.sound_feat_positive'] = d['sound_feat_positive'].numpy()
        # 4-channel image: only the first 3 channels are used (pretext_base.py:22)
        img4 = torch.cat([image, torch.ones(4, 1, 84, 84)], dim=1)
        d = m(img4, torch.from_numpy(pos), torch.from_numpy(neg))
        out['d.image_feat'] = d['image_feat'].numpy()
        out['d.sound_feat_negative'] = d['sound_feat_negative'].numpy()
        # RL reward (Envs/vec_env/vec_pretext_normalize.py:96-101): sum_d image_feat*goal_sound_feat
        out['d.reward'] = np.sum(d['image_feat'].numpy() * d['sound_feat_positive'].numpy(), axis=1)
    np.savez(os.path.join(HERE, "kuka_edge.npz"), **out)

    # --- lr schedule ---
    c = types.SimpleNamespace(pretextLRStep="step", pretextLRDecayEpoch=[10, 30, 50], pretextLRDecayGamma=0.2)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-4)
    sch = get_scheduler(c, opt)
    lrs = []
    for ep in range(60):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sch.step()
    np.savez(os.path.join(HERE, "lr_schedule.npz"), lrs=np.asarray(lrs, dtype=np.float64))
    print("done")


def make_reward_norm():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_rms", "/root/reference/Envs/vec_env/running_mean_std.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(77)
    n_env, steps, gamma, cliprew, eps = 8, 40, 0.99, 10.0, 1e-8
    rews = rng.normal(0.3, 1.5, size=(steps, n_env))
    rews[5] *= 40.0                                              # exercises the clip
    news = rng.random((steps, n_env)) < 0.08
    outs = {}
    for cliprew in (10.0, 1.5):                                  # the default, and one that actually clips
        rms = mod.RunningMeanStd(shape=())
        ret = np.zeros(n_env)
        out = np.zeros_like(rews)
        for t in range(steps):                                   # vec_pretext_normalize.py:52-59
            ret = ret * gamma + rews[t]
            rms.update(ret)
            out[t] = np.clip(rews[t] / np.sqrt(rms.var + eps), -cliprew, cliprew)
            ret[news[t]] = 0.
        outs[cliprew] = out
    np.savez(os.path.join(HERE, "reward_norm.npz"), rews=rews, news=news, out=outs[10.0], out_clip15=outs[1.5],
             mean=rms.mean, var=rms.var, count=rms.count)
    print("reward_norm.npz", rews.shape)


if __name__ == "__main__":
    main()
    make_reward_norm()
