"""Host logic of the RL-stage helper: streaming return normalisation against the fixture made with
the reference's RunningMeanStd, and (GPU) the intrinsic reward through the frozen HIP encoder."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_return_normalizer_matches_reference_stream(golden_dir):
    import var_amd
    fx = dict(np.load(os.path.join(golden_dir, "reward_norm.npz")))
    norm = var_amd.ReturnNormalizer(num_envs=fx["rews"].shape[1])
    out = np.stack([norm(fx["rews"][t], fx["news"][t]) for t in range(fx["rews"].shape[0])])
    assert np.array_equal(out, fx["out"])                         # same float64 arithmetic, same order
    assert norm.ret_rms.mean == fx["mean"] and norm.ret_rms.var == fx["var"] and norm.ret_rms.count == fx["count"]
    norm2 = var_amd.ReturnNormalizer(num_envs=fx["rews"].shape[1], cliprew=1.5)
    out2 = np.stack([norm2(fx["rews"][t], fx["news"][t]) for t in range(fx["rews"].shape[0])])
    assert np.array_equal(out2, fx["out_clip15"]) and np.max(np.abs(out2)) == 1.5 and np.mean(np.abs(out2) == 1.5) > 0.01


def test_running_mean_std_equals_batch_statistics():
    import var_amd
    rng = np.random.default_rng(0)
    x = rng.normal(2.0, 3.0, size=(1000, 4))
    rms = var_amd.RunningMeanStd(shape=(4,), epsilon=1e-12)
    for chunk in np.split(x, 10):
        rms.update(chunk)
    assert np.allclose(rms.mean, x.mean(0), atol=1e-9) and np.allclose(rms.var, x.var(0), atol=1e-9)


@pytest.mark.gpu
def test_intrinsic_reward_on_hip_encoder(golden_dir):
    """getEmbeddings + calcReward of the reference wrapper on the HIP forward at the RL batch (8 envs), incl. the
    cached-goal-sound protocol; expected values are the reference's own forward outputs (kuka_edge.npz)."""
    import torch
    import var_amd
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights.npz")))
    fx = dict(np.load(os.path.join(golden_dir, "kuka_edge.npz")))
    cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
    m = var_amd.VARPretextNet(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to("cuda")
    r = var_amd.IntrinsicReward(m)
    img, goal = fx["image"], fx["sound_positive"]
    image_feat, goal_feat, _ = r.embeddings(img, goal)
    env_rew = np.linspace(-1, 1, img.shape[0])
    total, dot, ss = r.reward(env_rew, image_feat, goal_feat)
    assert np.max(np.abs(dot - fx["d.reward"])) < 1e-4 and np.all(ss == 0)
    assert np.max(np.abs(total - (fx["d.reward"] + env_rew))) < 1e-4
    # later steps of the episode: the goal sound is not embedded again
    image_feat2, goal_feat2, _ = r.embeddings(img, None)
    assert np.array_equal(goal_feat2, goal_feat) and np.array_equal(image_feat2, image_feat)
    # the graph-captured latency path gives the same numbers (the image branch through the small-batch kernels,
    # include/var_hip.h: save_for_bwd = 2 -- equal within rounding; the sound branch bit for bit)
    r.capture(img.shape[0])
    f1, g1, dot1 = r.step(torch.from_numpy(img).cuda(), torch.from_numpy(goal).cuda())
    assert np.max(np.abs(f1.cpu().numpy() - image_feat)) < 1e-6 and np.array_equal(g1.cpu().numpy(), goal_feat)
    assert np.max(np.abs(dot1.cpu().numpy() - fx["d.reward"])) < 1e-4
    f2, g2, dot2 = r.step(torch.from_numpy(img).cuda(), None)
    assert np.array_equal(f2.cpu().numpy(), f1.cpu().numpy()) and np.array_equal(g2.cpu().numpy(), goal_feat)
    # (round 4: in that graph the image head's launch finishes the embeddings and takes the row dot itself, var_set_reward_dot:
    # same sums as heads_finish_kernel + row_dot_kernel)
    assert np.array_equal(dot2.cpu().numpy(), dot1.cpu().numpy())


@pytest.mark.gpu
def test_row_dot_is_calc_reward_and_rejects_bad_arguments():
    """var_row_dot = calcReward's torch.sum(a * b, dim=1) (vec_pretext_normalize.py:96-101) in one launch."""
    import torch
    import var_amd
    from var_amd._lib import Context, VarHipError, current_stream_handle, ptr
    c = Context.get(0)
    g = torch.Generator().manual_seed(3)
    for rows, dim in ((8, 3), (300, 3), (5, 64)):
        a, b = torch.randn(rows, dim, generator=g).cuda(), torch.randn(rows, dim, generator=g).cuda()
        out = torch.empty(rows, device="cuda")
        c.check(c.lib.var_row_dot(c.handle, current_stream_handle(), ptr(a), ptr(b), rows, dim, ptr(out)), "var_row_dot")
        want = (a.double() * b.double()).sum(1)
        np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-6 * dim)
    with pytest.raises(VarHipError):
        c.check(c.lib.var_row_dot(c.handle, current_stream_handle(), ptr(a), ptr(b), 5, 65, ptr(out)), "var_row_dot")
    with pytest.raises(VarHipError):
        c.check(c.lib.var_row_dot(c.handle, current_stream_handle(), None, ptr(b), 5, 3, ptr(out)), "var_row_dot")
