#!/usr/bin/env python3
"""Golden vectors for the RL actor-critic forward, made by IMPORTING the reference's
models.ppo.model.Policy with base 'arm_VAR' (models/RL/arm_RL_model.py:armNet_VAR) on CPU, Kuka configuration
(fourInARow/config.py:67-106: recurrent, 128 -> 512 GRU, action hidden 128, actionDim 2; kuka/env_config.py:37
robotStateDim 2; img_dim (3,96,96)).  gym is absent from the image; Policy only looks at the action space's class
name and shape, so the harness passes a stand-in object whose class is called Box.

The 3.3 M weights are not committed: the fixture stores the seed and per-tensor check values, which the oracle
(oracle/torch_oracle.py:ArmNetCPU) reproduces with the same constructor order; see tests/test_oracle_armnet.py.

armnet_b8.npz   inputs of RLNumEnvs = 8 environments (image u8, image_feat, robot_pose, goal_sound_feat, rnn_hxs, masks
                with one episode start), outputs of Policy.act(deterministic=True): value, action (= the mean),
                action_log_probs, rnn_hxs; plus base(...) actor features; a second step fed with the first's rnn_hxs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
SEED = 453


class Box:                                   # stand-in for gym.spaces.Box: Policy reads __class__.__name__ and .shape
    def __init__(self, n):
        self.shape = (n,)


def check_values(v):
    f = v.reshape(-1).astype(np.float64)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]])


def main():
    torch.set_num_threads(4)
    from models.ppo.model import Policy
    cfg = types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2)
    torch.manual_seed(SEED)
    ac = Policy(None, Box(2), base='arm_VAR', config=cfg,
                base_kwargs={'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512, 'actionHiddenSize': 128})
    ac.eval()
    out = {"seed": np.int64(SEED)}
    names = []
    for k, v in ac.state_dict().items():
        names.append(k)
        out["shape." + k] = np.asarray(v.shape, dtype=np.int64)
        out["check." + k] = check_values(v.numpy())
    out["names"] = np.asarray(names)
    rng = np.random.default_rng(21)
    B = 8
    img = rng.integers(0, 256, size=(B, 3, 96, 96), dtype=np.uint8)
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    obs = {
        'image': (torch.from_numpy(img) / 255.).float(),
        'image_feat': torch.from_numpy(unit(rng.standard_normal((B, 3)))),
        'robot_pose': torch.from_numpy(rng.uniform(-1, 1, size=(B, 2)).astype(np.float32)),
        'goal_sound_feat': torch.from_numpy(unit(rng.standard_normal((B, 3)))),
    }
    hxs = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32) * 0.3)
    masks = torch.ones(B, 1)
    masks[2] = 0.0                                            # env 2 starts a new episode
    out.update(image=img, image_feat=obs['image_feat'].numpy(), robot_pose=obs['robot_pose'].numpy(),
               goal_sound_feat=obs['goal_sound_feat'].numpy(), rnn_hxs=hxs.numpy(), masks=masks.numpy())
    with torch.no_grad():
        value, action, logp, hxs1 = ac.act(obs, hxs, masks, deterministic=True)
        _, feats, _, _ = ac.base(obs, hxs, masks)
        value2, action2, _, hxs2 = ac.act(obs, hxs1, torch.ones(B, 1), deterministic=True)
    out.update(value=value.numpy(), action=action.numpy(), action_log_probs=logp.numpy(), rnn_hxs_out=hxs1.numpy(),
               actor_features=feats.numpy(), value2=value2.numpy(), action2=action2.numpy(), rnn_hxs_out2=hxs2.numpy())
    np.savez_compressed(os.path.join(HERE, "armnet_b8.npz"), **out)
    print("params", sum(v.numel() for v in ac.state_dict().values()), "keys", len(names))
    print("value", value.numpy().ravel()[:4], "action", action.numpy()[:2], "logp", logp.numpy().ravel()[:2])


if __name__ == "__main__":
    main()
