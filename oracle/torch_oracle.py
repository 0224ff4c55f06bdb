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
    def __init__(self, state_dict=None, lr=1e-4, weight_decay=1e-6, margin=1.0, hw=84, dtype=torch.float32):
        """dtype=torch.float64: the same step on the same fp32 inputs and initial weights with every product, sum and the
        optimiser state in double -- the yardstick that tells the rounding drift of one fp32 implementation from another's."""
        self.model = KukaNetCPU(hw)
        if state_dict is not None:
            self.model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()})
        self.dtype = dtype
        self.model.to(dtype)
        self.model.train()
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        self.crit = torch.nn.TripletMarginLoss(margin=margin, p=2)

    def step(self, image_u8, pos, neg):
        image = (image_u8 / 255.).float().to(self.dtype)               # the fp32 quotient of dataset.py:67-68, then widened
        self.model.zero_grad()
        self.opt.zero_grad()
        a, p, n = self.model(image, pos.float().to(self.dtype), neg.float().to(self.dtype))
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


# ---- RL actor-critic ---------------------------------------------------------------
def _ortho(m, gain):
    nn.init.orthogonal_(m.weight.data, gain=gain)
    nn.init.constant_(m.bias.data, 0)
    return m


class _ArmBase(nn.Module):
    """models/RL/arm_RL_model.py:armNet_VAR (96x96 branch) on top of models/ppo/model.py:NNBase, restated from
    torch.nn layers with the reference's construction order, initialisers (orthogonal, gain sqrt 2; GRU weights
    orthogonal, biases 0) and its torch.rand shape probe, so that a seed reproduces the reference's weights."""

    def __init__(self, representation_dim=3, robot_state_dim=2, rin=128, rh=512, action_hidden=128):
        super().__init__()
        self.gru = nn.GRU(rin, rh)
        for name, p in self.gru.named_parameters():
            if 'bias' in name:
                nn.init.constant_(p, 0)
            elif 'weight' in name:
                nn.init.orthogonal_(p)
        self.imgCNN = nn.Sequential(
            nn.Conv2d(3, 32, 3, stride=1, padding=1), nn.ReLU(), nn.Conv2d(32, 32, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(32, 64, 3, stride=1, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 128, 3, stride=1, padding=1), nn.ReLU(), nn.Conv2d(128, 128, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(128, 256, 3, stride=2, padding=0), nn.ReLU(), nn.Conv2d(256, 128, 3, stride=1, padding=0), nn.ReLU(),
            nn.Flatten())
        self.imgCNN(torch.rand((1, 3, 96, 96)))               # get_layer_output_shape's probe (models/ppo/model.py:7-8)
        g = 2 ** 0.5
        lin = lambda i, o: _ortho(nn.Linear(i, o), g)         # noqa: E731
        self.motorMlp = nn.Sequential(lin(representation_dim + robot_state_dim, 256), nn.ReLU(), lin(256, 512), nn.ReLU(),
                                      lin(512, 256), nn.ReLU())
        self.cnnMlp = nn.Sequential(lin(1152, 512), nn.ReLU(), lin(512, 256), nn.ReLU())
        self.imgMotorMlp = nn.Sequential(lin(256, 256), nn.ReLU(), lin(256, rin), nn.ReLU())
        self.imgMotorMlp2 = nn.Sequential(lin(rh, 256), nn.ReLU())
        self.soundMlp = nn.Sequential(lin(representation_dim, 128), nn.ReLU(), lin(128, 256), nn.ReLU(), lin(256, 256), nn.ReLU())
        self.fusionMlp = nn.Sequential(lin(256, 512), nn.ReLU(), lin(512, 256), nn.ReLU())
        self.mlp_all = nn.Sequential(lin(256, 256), nn.ReLU(), lin(256, 128), nn.ReLU())
        self.actor = nn.Sequential(lin(128, 128), nn.ReLU(), lin(128, action_hidden), nn.ReLU())
        self.critic = nn.Sequential(lin(128, 128), nn.ReLU(), lin(128, 128), nn.ReLU())
        self.critic_linear = lin(128, 1)

    def forward(self, obs, rnn_hxs, masks):
        image_flatten = self.cnnMlp(self.imgCNN(obs['image']))
        motor = self.motorMlp(torch.cat([obs['image_feat'], obs['robot_pose']], dim=1))
        image_motor = self.imgMotorMlp(image_flatten + motor)
        x, h = self.gru(image_motor.unsqueeze(0), (rnn_hxs * masks).unsqueeze(0))     # models/ppo/model.py:118-121
        image_motor, rnn_hxs = x.squeeze(0), h.squeeze(0)
        fusion = self.fusionMlp(self.soundMlp(obs['goal_sound_feat']) + image_flatten)
        x = self.mlp_all(fusion + self.imgMotorMlp2(image_motor))
        return self.critic_linear(self.critic(x)), self.actor(x), rnn_hxs


class _AddBias(nn.Module):
    def __init__(self, n):
        super().__init__()
        self._bias = nn.Parameter(torch.zeros(n).unsqueeze(1))


class _DiagGaussian(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.fc_mean = _ortho(nn.Linear(num_inputs, num_outputs), 1)
        self.logstd = _AddBias(num_outputs)


class ArmNetCPU(nn.Module):
    """Policy(base='arm_VAR') of models/ppo/model.py:15-69 with a Box action space: base + DiagGaussian; act() in
    deterministic mode returns (value, mean action, log-prob of the mean, rnn_hxs)."""

    def __init__(self, num_actions=2):
        super().__init__()
        self.base = _ArmBase()
        self.dist = _DiagGaussian(128, num_actions)

    def act_deterministic(self, obs, rnn_hxs, masks):
        value, feats, rnn_hxs = self.base(obs, rnn_hxs, masks)
        mean = self.dist.fc_mean(feats)
        std = self.dist.logstd._bias.t().view(1, -1).expand_as(mean).exp()
        logp = torch.distributions.Normal(mean, std).log_prob(mean).sum(-1, keepdim=True)
        return value, mean, logp, rnn_hxs, feats


def armnet_seeded(seed=453):
    torch.manual_seed(seed)
    m = ArmNetCPU()
    m.eval()
    return m


# ---- in-batch-negatives contrastive head (extension; csrc/inbatch.hip) ---------------------------------------------
def inbatch_contrastive_loss(anchor, cand, target, tau=0.1, inv_count=None):
    """sum_i [logsumexp_j(-d_ij/tau) + d_{i,target[i]}/tau] * inv_count with d_ij = ||a_i - c_j + 1e-6||_2
    (torch.nn.functional.pairwise_distance, the convention TripletMarginLoss uses); inv_count = 1/B by default, i.e.
    cross-entropy over negative distances."""
    d = F.pairwise_distance(anchor[:, None, :], cand[None, :, :], p=2, eps=1e-6)
    loss = F.cross_entropy(-d / tau, target, reduction="sum")
    return loss * (1.0 / anchor.shape[0] if inv_count is None else inv_count)
