"""The PyTorch-CPU restatement (oracle/torch_oracle.py, bench.py's cpu_baseline) against the
vectors the reference produced."""
import os

import numpy as np
import torch

from oracle.torch_oracle import CPUTrainer


def test_cpu_trainer_matches_reference_adam_trajectory(golden_dir):
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights.npz")))
    fx = dict(np.load(os.path.join(golden_dir, "kuka_adam.npz")))
    torch.set_num_threads(1)
    tr = CPUTrainer(sd)
    for s in range(3):
        loss = tr.step(torch.from_numpy(fx[f'image{s}']), torch.from_numpy(fx[f'pos{s}']), torch.from_numpy(fx[f'neg{s}']))
        assert abs(loss - float(fx['losses'][s])) < 1e-6
    got = tr.model.state_dict()
    for k, v in got.items():
        assert np.max(np.abs(v.numpy() - fx['step3.' + k])) < 1e-6, k
