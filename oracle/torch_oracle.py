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


class IthorNetCPU(nn.Module):
    """iTHOR VARPretextNet restated from torch.nn layers (models/pretext/ai2thor_pretext_model.py:5-58): image CNN of
    stride-1 3x3 convolutions with 2x2 max pools, sound CNN of three wide stride-2 convolutions feeding a
    bidirectional GRU(448 -> 512) whose two final hidden states are concatenated, Linear heads, F.normalize.
    Modules are created in the reference's order (imgBranch, rnn, cnn, imgTriplet, soundTriplet) so that
    `torch.manual_seed(977); IthorNetCPU()` draws the reference's initial weights (checked against
    tests/golden/ithor_h96.npz by tests/test_oracle_ithor.py)."""

    def __init__(self):
        super().__init__()
        self.imgBranch = nn.Sequential(
            nn.Conv2d(3, 32, 3, stride=1, padding=1), nn.ReLU(), nn.Conv2d(32, 32, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2), nn.Conv2d(32, 64, 3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 128, 3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(2, stride=2),
            nn.Conv2d(128, 128, 3, stride=2, padding=1), nn.ReLU(), nn.Flatten())
        self.rnn = nn.GRU(input_size=64 * 7, hidden_size=512, batch_first=True, bidirectional=True)
        self.cnn = nn.Sequential(
            nn.Conv2d(1, 64, (11, 11), stride=(2, 2), padding=(5, 5)), nn.ReLU(),
            nn.Conv2d(64, 64, (11, 5), stride=(2, 2), padding=(5, 5)), nn.ReLU(),
            nn.Conv2d(64, 64, (7, 3), stride=(2, 2), padding=(1, 1)), nn.ReLU())
        self.imgTriplet = nn.Sequential(nn.Linear(128 * 9, 128), nn.ReLU(), nn.Linear(128, 3))
        self.soundTriplet = nn.Sequential(nn.Linear(2 * 512, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                          nn.Linear(64, 3))

    def sound_raw(self, sound):
        c = self.cnn(sound)
        seq = torch.reshape(torch.transpose(c, 1, 2), (-1, c.shape[2], 64 * c.shape[3]))
        _, h = self.rnn(seq)
        return torch.cat((h[0], h[1]), dim=1)

    def forward(self, image, sound_positive, sound_negative, raw=False):
        image_raw = self.imgBranch(image[:, :3])
        image_feat = F.normalize(self.imgTriplet(image_raw), p=2, dim=1)
        pos_raw = self.sound_raw(sound_positive)
        sp = F.normalize(self.soundTriplet(pos_raw), p=2, dim=1)
        sn = F.normalize(self.soundTriplet(self.sound_raw(sound_negative)), p=2, dim=1)
        if raw:
            return image_feat, sp, sn, image_raw, pos_raw
        return image_feat, sp, sn


def ithor_seeded(seed=977):
    torch.manual_seed(seed)
    m = IthorNetCPU()
    m.train()
    return m
