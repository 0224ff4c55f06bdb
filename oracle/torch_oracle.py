"""ORACLE (test infrastructure / cpu_baseline only -- never imported by the product path).

PyTorch-CPU restatement of the reference's training step with the Kuka architecture
re-declared from torch.nn layers:
    models/pretext/arm_pretext_model.py:9-56   (layers)
    models/pretext/pretext_base.py:10-41       (routing + F.normalize)
    VAR/pretext_VAR.py:33-39,55-70             (Adam, TripletMarginLoss, step body)
This is what the reference itself executes on a CPU-only host (torch.nn -> oneDNN), so it is the
`cpu_baseline` that bench.py times beside the HIP path.  Pinned by tests/test_oracle_torch.py
against the golden vectors."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class KukaNetCPU(nn.Module):
    def __init__(self, hw=84):
        super().__init__()
        self.imgBranch = nn.Sequential(
            nn.Conv2d(3, 32, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(32, 32, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(32, 64, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU(), nn.Flatten())
        self.soundCNN = nn.Sequential(
            nn.Conv2d(1, 32, (5, 40), stride=(2, 1)), nn.ReLU(), nn.Conv2d(32, 32, (3, 1), stride=(2, 1)), nn.ReLU(),
            nn.Conv2d(32, 32, (3, 1), stride=(2, 1)), nn.ReLU(), nn.Conv2d(32, 32, (3, 1), stride=(2, 1)), nn.ReLU(),
            nn.Flatten())
        self.imgTriplet = nn.Sequential(nn.Linear(576, 128), nn.ReLU(), nn.Linear(128, 3))
        self.soundTriplet = nn.Sequential(nn.Linear(160, 128), nn.ReLU(), nn.Linear(128, 3))

    def forward(self, image, sound_positive, sound_negative):
        image_feat = F.normalize(self.imgTriplet(self.imgBranch(image[:, :3])), p=2, dim=1)
        sp = F.normalize(self.soundTriplet(self.soundCNN(sound_positive)), p=2, dim=1)
        sn = F.normalize(self.soundTriplet(self.soundCNN(sound_negative)), p=2, dim=1)
        return image_feat, sp, sn


class CPUTrainer:
    def __init__(self, state_dict=None, lr=1e-4, weight_decay=1e-6, margin=1.0):
        self.model = KukaNetCPU()
        if state_dict is not None:
            self.model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()})
        self.model.train()
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        self.crit = torch.nn.TripletMarginLoss(margin=margin, p=2)

    def step(self, image_u8, pos, neg):
        image = (image_u8 / 255.).float()
        self.model.zero_grad()
        self.opt.zero_grad()
        a, p, n = self.model(image, pos.float(), neg.float())
        loss = self.crit(a, p, n)
        loss.backward()
        self.opt.step()
        return loss.item()
