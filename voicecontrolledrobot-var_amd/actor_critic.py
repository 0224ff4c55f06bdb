"""Drop-in for the acting half of the reference's `Policy(base='arm_VAR')` (models/ppo/model.py:15-69 over
models/RL/arm_RL_model.py:armNet_VAR, Kuka configuration): same constructor arguments, module tree and state_dict
keys (63 tensors -- the authors' RL checkpoints load unchanged), `act` / `get_value` / `is_recurrent` /
`recurrent_hidden_state_size`.  The forward (8 convolutions, 3 max pools, the MLPs, one GRU step, value head, actor
trunk, DiagGaussian mean) is ONE C-ABI call, var_armnet_forward (csrc/armnet.hip); sampling and log-probabilities
(a handful of flops on (B,2) tensors) use torch.distributions exactly as the reference's FixedNormal does.
Inference only: evaluate_actions (the PPO update, models/ppo/algo/ppo.py) stays in PyTorch.  GPU only."""
import numpy as np
import torch
import torch.nn as nn

from ._lib import Context, VarHipError, current_stream_handle, ptr


def _ortho(m, gain):
    nn.init.orthogonal_(m.weight.data, gain=gain)
    nn.init.constant_(m.bias.data, 0)
    return m


class _Base(nn.Module):
    """Parameter container with armNet_VAR's attribute names, construction order and initialisers."""

    def __init__(self, config, recurrent, rin, rh, action_hidden):
        super().__init__()
        self.config = config
        self._recurrent, self._recurrent_size, self._action_hidden_size = recurrent, rh, action_hidden
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
        torch.rand((1, *config.img_dim))                      # the reference's shape probe draws here
        self.imgCNN_outputShape = torch.Size((1, 1152))
        g = float(np.sqrt(2))
        lin = lambda i, o: _ortho(nn.Linear(i, o), g)         # noqa: E731
        self.motorMlp = nn.Sequential(lin(config.representationDim + config.robotStateDim, 256), nn.ReLU(),
                                      lin(256, 512), nn.ReLU(), lin(512, 256), nn.ReLU())
        self.cnnMlp = nn.Sequential(lin(1152, 512), nn.ReLU(), lin(512, 256), nn.ReLU())
        self.imgMotorMlp = nn.Sequential(lin(256, 256), nn.ReLU(), lin(256, rin), nn.ReLU())
        self.imgMotorMlp2 = nn.Sequential(lin(rh, 256), nn.ReLU())
        self.soundMlp = nn.Sequential(lin(config.representationDim, 128), nn.ReLU(), lin(128, 256), nn.ReLU(),
                                      lin(256, 256), nn.ReLU())
        self.fusionMlp = nn.Sequential(lin(256, 512), nn.ReLU(), lin(512, 256), nn.ReLU())
        self.mlp_all = nn.Sequential(lin(256, 256), nn.ReLU(), lin(256, 128), nn.ReLU())
        self.actor = nn.Sequential(lin(128, 128), nn.ReLU(), lin(128, action_hidden), nn.ReLU())
        self.critic = nn.Sequential(lin(128, 128), nn.ReLU(), lin(128, 128), nn.ReLU())
        self.critic_linear = lin(128, 1)


class _AddBias(nn.Module):
    def __init__(self, n):
        super().__init__()
        self._bias = nn.Parameter(torch.zeros(n).unsqueeze(1))


class _DiagGaussian(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.fc_mean = _ortho(nn.Linear(num_inputs, num_outputs), 1)
        self.logstd = _AddBias(num_outputs)


class ArmNetPolicy(nn.Module):
    def __init__(self, obs_shape, action_space, config=None, base='arm_VAR', base_kwargs=None):
        super().__init__()
        kw = dict(recurrent=False, recurrentInputSize=128, recurrentSize=128, actionHiddenSize=128)
        kw.update(base_kwargs or {})
        if base != 'arm_VAR' or action_space.__class__.__name__ != "Box":
            raise NotImplementedError("HIP policy: base 'arm_VAR' with a Box action space")
        n_act = int(action_space.shape[0])
        if (tuple(config.img_dim) != (3, 96, 96) or config.representationDim != 3 or config.robotStateDim != 2
                or not kw['recurrent'] or kw['recurrentInputSize'] != 128 or kw['recurrentSize'] != 512
                or kw['actionHiddenSize'] != 128 or n_act != 2):
            raise VarHipError("HIP armNet_VAR supports the Kuka configuration: img_dim (3,96,96), representationDim 3, "
                              "robotStateDim 2, recurrent 128 -> 512, actionHiddenSize 128, 2 actions")
        self.base = _Base(config, True, 128, 512, 128)
        self.dist = _DiagGaussian(128, n_act)
        self._flat = None
        self._plan = 0
        self._flatten_params()

    @property
    def is_recurrent(self):
        return True

    @property
    def recurrent_hidden_state_size(self):
        return 512

    def _flatten_params(self):
        params = [p for _, p in self.named_parameters()]
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
        o = 0
        for p in params:
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
            o += p.numel()
        self._flat = flat

    def _arena_intact(self):
        o = self._flat.data_ptr()
        for _, p in self.named_parameters():
            if p.data_ptr() != o or p.dtype != torch.float32:
                return False
            o += 4 * p.numel()
        return True

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flatten_params()
        return r

    def forward(self, inputs, rnn_hxs, masks):
        raise NotImplementedError                             # as the reference (models/ppo/model.py:55-56)

    def _base_forward(self, inputs, rnn_hxs, masks):
        if not self._arena_intact():
            self._flatten_params()
        flat = self._flat
        if not flat.is_cuda:
            raise VarHipError("ArmNetPolicy runs on the GPU only: call .to('cuda') (no CPU fallback)")
        c = Context.get(flat.device.index)
        if flat.numel() != c.lib.var_armnet_param_count():
            raise VarHipError("parameter arena does not match var_armnet_param_count()")
        image = inputs['image']
        B = image.shape[0]
        if self._plan < B:
            c.check(c.lib.var_armnet_plan(c.handle, int(B)), "var_armnet_plan")
            self._plan = B
        f32 = lambda t, shape: self._prep(t, shape)           # noqa: E731
        if not image.is_cuda:
            raise VarHipError("inputs must be CUDA tensors (no CPU fallback)")
        if image.dtype != torch.uint8:
            image = image.float()
        image = image.reshape(B, -1, 96, 96).contiguous()
        feat, pose = f32(inputs['image_feat'], (B, 3)), f32(inputs['robot_pose'], (B, 2))
        goal = f32(inputs['goal_sound_feat'], (B, 3))
        hxs, m = f32(rnn_hxs, (B, 512)), f32(masks, (B, 1))
        dev = flat.device
        value = torch.empty((B, 1), dtype=torch.float32, device=dev)
        feats = torch.empty((B, 128), dtype=torch.float32, device=dev)
        mean = torch.empty((B, 2), dtype=torch.float32, device=dev)
        hout = torch.empty((B, 512), dtype=torch.float32, device=dev)
        c.check(c.lib.var_armnet_forward(c.handle, current_stream_handle(), ptr(flat), ptr(image),
                                         int(image.dtype == torch.uint8), image.stride(0), ptr(feat), ptr(pose), ptr(goal),
                                         ptr(hxs), ptr(m), B, ptr(value), ptr(feats), ptr(mean), ptr(hout)),
                "var_armnet_forward")
        return value, feats, mean, hout

    def chain_status(self):
        """Status of the small-batch (B <= 8) MLP chain launch, a persistent kernel that needs its 128 workgroups resident at
        once: 1 = the most recent forward timed out (its outputs are NaN), 0x40000001 = an earlier one did since the last
        clear_chain_status(), 0 = never.  Blocking.  (The reference's Policy.act, models/ppo/model.py:57-69, cannot fail;
        a caller that shares the GPU checks this after a NaN value or once per rollout.)"""
        import ctypes
        c = Context.get(self._flat.device.index)
        w = ctypes.c_uint(0)
        c.check(c.lib.var_armnet_status(c.handle, ctypes.byref(w)), "var_armnet_status")
        return int(w.value)

    def clear_chain_status(self):
        c = Context.get(self._flat.device.index)
        c.check(c.lib.var_armnet_clear_status(c.handle), "var_armnet_clear_status")

    @staticmethod
    def _prep(t, shape):
        if not t.is_cuda:
            raise VarHipError("inputs must be CUDA tensors (no CPU fallback)")
        t = t.float().contiguous()
        if tuple(t.shape) != tuple(shape):
            raise VarHipError(f"expected shape {shape}, got {tuple(t.shape)}")
        return t

    def _normal(self, mean):
        std = self.dist.logstd._bias.t().view(1, -1).expand_as(mean).exp()
        return torch.distributions.Normal(mean, std)

    @torch.no_grad()
    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        """models/ppo/model.py:57-69: (value, action, action_log_probs, rnn_hxs)."""
        value, _feats, mean, rnn_hxs = self._base_forward(inputs, rnn_hxs, masks)
        dist = self._normal(mean)
        action = mean if deterministic else dist.sample()
        return value, action, dist.log_prob(action).sum(-1, keepdim=True), rnn_hxs

    @torch.no_grad()
    def get_value(self, inputs, rnn_hxs, masks):
        return self._base_forward(inputs, rnn_hxs, masks)[0]

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        raise NotImplementedError("the PPO update (models/ppo/algo/ppo.py) stays in PyTorch: load this state_dict into "
                                  "the reference Policy for training")
