"""The iTHOR oracle (oracle/torch_oracle.py:IthorNetCPU) against the fixture made from the reference class
(tests/golden/make_golden_ithor.py): seed-regenerated weights, forward dict, loss, gradients, two Adam steps."""
import os

import numpy as np
import torch

from oracle.torch_oracle import ithor_seeded

G = os.path.join(os.path.dirname(__file__), "golden", "ithor_h96.npz")


def check_values(v):
    f = v.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]])


def test_ithor_oracle_matches_reference_fixture():
    torch.set_num_threads(4)
    g = np.load(G)
    m = ithor_seeded(int(g["seed"]))
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["names"]]
    for k, v in sd.items():
        assert tuple(g["shape." + k]) == tuple(v.shape), k
        np.testing.assert_array_equal(check_values(v.numpy()), g["check." + k], err_msg=k)
    stride = int(g["stride"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    crit = torch.nn.TripletMarginLoss(margin=1.0, p=2)
    image = (torch.from_numpy(g["image"]) / 255.).float()
    pos, neg = torch.from_numpy(g["sound_positive"]), torch.from_numpy(g["sound_negative"])
    for step in range(2):
        opt.zero_grad()
        a, p, n, iraw, praw = m(image, pos, neg, raw=True)
        loss = crit(a, p, n)
        loss.backward()
        assert abs(loss.item() - float(g["losses"][step])) < 1e-6
        if step == 0:
            for name, val in (("image_feat", a), ("sound_feat_positive", p), ("sound_feat_negative", n),
                              ("image_feat_raw", iraw), ("pos_sound_raw", praw)):
                np.testing.assert_allclose(val.detach().numpy(), g[name], rtol=1e-5, atol=1e-6, err_msg=name)
            for k, q in m.named_parameters():
                gr = q.grad.numpy().reshape(-1)
                np.testing.assert_allclose(gr[::stride], g["gsamp." + k], rtol=1e-4, atol=1e-7, err_msg=k)
                assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - float(g["gnorm." + k])) <= 1e-5 * float(g["gnorm." + k]) + 1e-9
        opt.step()
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.numpy().reshape(-1)[::stride], g["adam2." + k], rtol=1e-5, atol=1e-7, err_msg=k)
