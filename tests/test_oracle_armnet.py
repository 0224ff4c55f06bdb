"""The actor-critic oracle (oracle/torch_oracle.py:ArmNetCPU) against the fixture made from the reference's Policy
(tests/golden/make_golden_armnet.py): seed-regenerated weights, value / action / log-prob / hidden state."""
import os

import numpy as np
import torch

from oracle.torch_oracle import armnet_seeded

G = os.path.join(os.path.dirname(__file__), "golden", "armnet_b8.npz")


def test_armnet_oracle_matches_reference_fixture():
    g = np.load(G)
    m = armnet_seeded(int(g["seed"]))
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["names"]]
    for k, v in sd.items():
        f = v.numpy().reshape(-1).astype(np.float64)
        assert tuple(g["shape." + k]) == tuple(v.shape), k
        # orthogonal_ goes through a LAPACK QR whose last bits depend on the BLAS build / thread count: close, not equal
        np.testing.assert_allclose(np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]]), g["check." + k], rtol=1e-5,
                                   atol=1e-6, err_msg=k)
    obs = {'image': (torch.from_numpy(g['image']) / 255.).float(), 'image_feat': torch.from_numpy(g['image_feat']),
           'robot_pose': torch.from_numpy(g['robot_pose']), 'goal_sound_feat': torch.from_numpy(g['goal_sound_feat'])}
    with torch.no_grad():
        v, a, lp, h, f = m.act_deterministic(obs, torch.from_numpy(g['rnn_hxs']), torch.from_numpy(g['masks']))
        v2, a2, _, h2, _ = m.act_deterministic(obs, h, torch.ones(8, 1))
    for got, name in ((v, 'value'), (a, 'action'), (lp, 'action_log_probs'), (h, 'rnn_hxs_out'), (f, 'actor_features'),
                      (v2, 'value2'), (a2, 'action2'), (h2, 'rnn_hxs_out2')):
        np.testing.assert_allclose(got.numpy(), g[name], rtol=1e-4, atol=1e-5, err_msg=name)
