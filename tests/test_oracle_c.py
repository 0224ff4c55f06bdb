"""The C oracle (oracle/var_oracle.c) against vectors computed by the reference itself.

Fixtures: tests/golden/*.npz, written by tests/golden/make_golden.py, which imports
models.pretext.arm_pretext_model.VARPretextNet from /root/reference on CPU."""
import os

import numpy as np
import pytest

from oracle import var_oracle as orc


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize("wfile,ffile", [("kuka_weights.npz", "kuka_h84.npz"),
                                         ("kuka_weights.npz", "kuka_h96.npz"),
                                         ("kuka_weights2.npz", "kuka_h84_w2.npz")])
def test_forward_loss_grads(golden_dir, wfile, ffile):
    sd = load(golden_dir, wfile)
    fx = load(golden_dir, ffile)
    P = orc.flatten_params(sd)
    out = orc.forward(P, fx['image'], fx['sound_positive'], fx['sound_negative'])
    for k in ('image_feat', 'sound_feat_positive', 'sound_feat_negative', 'image_feat_raw', 'pos_sound_raw'):
        assert np.max(np.abs(out[k] - fx[k])) < 1e-5, k
    loss, G, (a, p, n) = orc.loss_grad(P, fx['image'], fx['sound_positive'], fx['sound_negative'])
    assert abs(loss - float(fx['loss'])) < 1e-6
    g = orc.unflatten_params(G)
    for k, _ in orc.PARAM_SPECS:
        ref = fx['grad.' + k]
        assert rel_err(g[k], ref) < 2e-4, (k, rel_err(g[k], ref))
    # f32 image input path (image/255 done by the caller, dataset.py:67-68)
    out2 = orc.forward(P, fx['image'].astype(np.float32) / np.float32(255.), fx['sound_positive'], None)
    assert np.array_equal(out2['image_feat'], out['image_feat'])
    assert out2['sound_feat_negative'] is None


def test_adam_trajectory(golden_dir):
    sd = load(golden_dir, "kuka_weights.npz")
    fx = load(golden_dir, "kuka_adam.npz")
    P = orc.flatten_params(sd).copy()
    m = np.zeros_like(P)
    v = np.zeros_like(P)
    for s in range(3):
        loss, G, _ = orc.loss_grad(P, fx[f'image{s}'], fx[f'pos{s}'], fx[f'neg{s}'])
        assert abs(loss - float(fx['losses'][s])) < 2e-6
        orc.adam(P, G, m, v, s + 1)
        if s in (0, 2):
            ref = orc.flatten_params({k: fx[f'step{s + 1}.' + k] for k, _ in orc.PARAM_SPECS})
            # Adam moves every weight by ~lr=1e-4 per step; compare the displacement
            d_ref = ref - orc.flatten_params(sd)
            d_got = P - orc.flatten_params(sd)
            # Adam normalises g by sqrt(v)+1e-8, so weights whose gradient is ~1e-8..1e-5
            # (a ReLU gate flipping on a rounding difference) move by O(lr) either way:
            # require 99.9% of the arena within 2e-6 and everything within the s+1 steps' reach.
            diff = np.abs(P - ref)
            assert np.mean(diff < 2e-6) > 0.999
            assert np.max(diff) < 1.05e-4 * (s + 1)
            assert np.max(np.abs(d_ref)) < 1.05e-4 * (s + 1) and np.max(np.abs(d_got)) < 1.05e-4 * (s + 1)


def test_edge_behaviours(golden_dir):
    sd = load(golden_dir, "kuka_weights.npz")
    fx = load(golden_dir, "kuka_edge.npz")
    P = orc.flatten_params(sd)
    a = orc.forward(P, fx['image'], fx['sound_positive'], None)
    assert np.max(np.abs(a['image_feat'] - fx['a.image_feat'])) < 1e-5
    assert np.max(np.abs(a['sound_feat_positive'] - fx['a.sound_feat_positive'])) < 1e-5
    assert np.max(np.abs(a['pos_sound_raw'] - fx['a.pos_sound_raw'])) < 1e-5
    c = orc.forward(P, None, fx['sound_negative'], None)
    assert c['image_feat'] is None
    assert np.max(np.abs(c['sound_feat_positive'] - fx['c.sound_feat_positive'])) < 1e-5
    d = orc.forward(P, fx['image'], fx['sound_positive'], fx['sound_negative'])
    assert np.max(np.abs(d['sound_feat_negative'] - fx['d.sound_feat_negative'])) < 1e-5
    reward = np.sum(d['image_feat'] * d['sound_feat_positive'], axis=1)
    assert np.max(np.abs(reward - fx['d.reward'])) < 1e-5


def test_lr_schedule(golden_dir):
    lrs = load(golden_dir, "lr_schedule.npz")['lrs']
    for ep, lr in enumerate(lrs):
        assert abs(orc.multistep_lr(1e-4, [10, 30, 50], 0.2, ep) - lr) < 1e-15


def test_triplet_formula():
    rng = np.random.default_rng(0)
    a, p, n = (rng.standard_normal((16, 3)).astype(np.float32) for _ in range(3))
    loss, ga, gp, gn = orc.triplet(a, p, n)
    dp = np.linalg.norm(a - p + 1e-6, axis=1)
    dn = np.linalg.norm(a - n + 1e-6, axis=1)
    assert abs(loss - np.mean(np.maximum(dp - dn + 1.0, 0))) < 1e-6
    assert np.allclose(ga + gp + gn, 0, atol=1e-7)
