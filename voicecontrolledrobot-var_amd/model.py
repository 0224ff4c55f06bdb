"""Drop-in for the reference's `config.pretextModel`: same constructor, attribute names,
state_dict layout and forward() contract as models/pretext/arm_pretext_model.py:VARPretextNet
(+ models/pretext/pretext_base.py:PretextNetBase.VAR_forward); the arithmetic is the HIP library."""
import numpy as np
import torch
import torch.nn as nn

from ._lib import Context, VarHipError, current_stream_handle, ptr
from .layout import N_PARAMS, PARAM_OFFSETS, PARAM_SPECS


class Flatten(nn.Module):              # utils.py:9-11 (kept so module indices/state_dict keys match)
    def forward(self, x):
        return x.view(x.size(0), -1)


def _conv_out(h):
    return (h - 1) // 2 + 1


def _encoder_forward(module, need_grad, image, pos, neg):
    """var_arm_encoder_fwd on this model's packed weights; returns (outs, B, generation of the saved forward)."""
    flat = module._flat
    dev = flat.device
    c = Context.get(dev.index)
    ref = image if image is not None else (pos if pos is not None else neg)
    B = ref.shape[0]
    H = module.config.img_dim[1]
    c.ensure_plan(B, H)
    stream = current_stream_handle()
    # The eager module path re-packs on EVERY forward (one ~9 us kernel): torch's version counters do not see edits made
    # through `.data` (p.data.mul_(), nn.init on m.weight.data), and a stale packed image is a silent wrong answer.  The
    # latency paths -- VARTrainer's captured steps, IntrinsicReward's graphs -- own their image and never come through here.
    module.hip_weights(c, force=True)
    mk = lambda n: torch.empty((B, n), dtype=torch.float32, device=dev)
    image_feat = mk(3) if image is not None else None
    image_raw = mk(576) if image is not None else None
    pos_feat = mk(3) if pos is not None else None
    pos_raw = mk(160) if pos is not None else None
    neg_feat = mk(3) if neg is not None else None
    is_u8 = image is not None and image.dtype == torch.uint8
    bstride = 0 if image is None else image.stride(0)
    c.check(c.lib.var_arm_encoder_fwd(c.handle, stream, ptr(flat), ptr(image), int(is_u8), bstride,
                                      ptr(pos), ptr(neg), B, H, ptr(image_feat), ptr(pos_feat),
                                      ptr(neg_feat), ptr(image_raw), ptr(pos_raw), int(need_grad)),
            "var_arm_encoder_fwd")
    gen = c.lib.var_saved_generation(c.handle) if need_grad else 0
    return (image_feat, pos_feat, neg_feat, image_raw, pos_raw), B, gen


class _EncoderFn(torch.autograd.Function):
    """forward/backward of the whole encoder through the C ABI (var_arm_encoder_fwd/_bwd)."""

    @staticmethod
    def forward(ctx, module, need_grad, image, pos, neg, *params):
        outs, B, gen = _encoder_forward(module, need_grad, image, pos, neg)
        dev = module._flat.device
        ctx.module = module
        ctx.keep = (image, pos, neg)          # the C side re-reads the inputs in backward
        ctx.B = B
        ctx.gen = gen
        ctx.present = [o is not None for o in outs]
        dummy = torch.zeros(0, device=dev)
        res = tuple(o if o is not None else dummy for o in outs)
        ctx.mark_non_differentiable(*[r for r, p in zip(res[3:], ctx.present[3:])])
        return res

    @staticmethod
    def backward(ctx, g_if, g_pf, g_nf, g_ir, g_pr):
        module = ctx.module
        flat = module._flat
        c = Context.get(flat.device.index)
        if not ctx.gen or c.lib.var_saved_generation(c.handle) != ctx.gen:
            raise VarHipError("backward of a forward whose activations are gone: the device context keeps ONE saved "
                              "forward (a later forward of this or another model overwrote it, or it ran under "
                              "no_grad) -- run forward and backward back to back")
        module.hip_weights(c)
        # a fresh buffer per backward: autograd may keep the returned views as .grad (AccumulateGrad steals them)
        gflat = torch.empty(N_PARAMS, dtype=torch.float32, device=flat.device)
        gs = []
        for g, present in zip((g_if, g_pf, g_nf), ctx.present[:3]):
            gs.append(g.contiguous().float() if (present and g is not None) else None)
        c.check(c.lib.var_arm_encoder_bwd(c.handle, current_stream_handle(), ptr(flat), ptr(gs[0]), ptr(gs[1]),
                                          ptr(gs[2]), ptr(gflat)), "var_arm_encoder_bwd")
        grads = [gflat[o:o + int(np.prod(s))].view(s) for o, (_, s) in zip(PARAM_OFFSETS, PARAM_SPECS)]
        return (None, None, None, None, None, *grads)


class VARPretextNet(nn.Module):
    """Kuka VAR encoder (image CNN + MFCC CNN + two triplet heads), HIP-backed.

    ctor(config) reads config.img_dim, config.sound_dim, config.representationDim exactly as the
    reference does (arm_pretext_model.py:45-56); forward(image, sound_positive, sound_negative,
    is_train=False) returns the same 7-key dict as pretext_base.py:37-40."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        if tuple(config.sound_dim) != (1, 100, 40) or config.representationDim != 3 \
                or config.img_dim[0] != 3 or config.img_dim[1] != config.img_dim[2] \
                or config.img_dim[1] not in (84, 96):
            raise VarHipError("HIP VARPretextNet supports img_dim (3,84,84)|(3,96,96), sound_dim (1,100,40), "
                              f"representationDim 3; got {config.img_dim} {config.sound_dim} {config.representationDim}")
        self.cached_sound = None
        # Same modules, same construction order and the same RNG draws as the reference ctor
        # (buildCNN, buildSoundBranch, then the two torch.rand shape probes between the heads),
        # so that a given torch.manual_seed yields the reference's initial weights.
        self.imgBranch = nn.Sequential(
            nn.Conv2d(3, 32, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(32, 32, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(32, 64, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU(),
            nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU(),
            Flatten())
        self.soundCNN = nn.Sequential(
            nn.Conv2d(1, 32, (5, 40), stride=(2, 1)), nn.ReLU(),
            nn.Conv2d(32, 32, (3, 1), stride=(2, 1)), nn.ReLU(),
            nn.Conv2d(32, 32, (3, 1), stride=(2, 1)), nn.ReLU(),
            nn.Conv2d(32, 32, (3, 1), stride=(2, 1)), nn.ReLU(),
            Flatten())
        torch.rand((1, *config.img_dim))                 # get_layer_output_shape's probe (models/ppo/model.py:7-8)
        h = config.img_dim[1]
        for _ in range(5):
            h = _conv_out(h)
        self.imgCNN_outputShape = torch.Size((1, 64 * h * h))
        self.imgTriplet = nn.Sequential(nn.Linear(64 * h * h, 128), nn.ReLU(),
                                        nn.Linear(128, config.representationDim))
        torch.rand(*config.sound_dim)                    # soundBranch shape probe (arm_pretext_model.py:51)
        self.soundBranch_outputShape = torch.Size((1, 160))
        self.soundTriplet = nn.Sequential(nn.Linear(160, 128), nn.ReLU(),
                                          nn.Linear(128, config.representationDim))
        self._flat = None
        self._weights = None
        self._flatten_params()

    # ---- flat parameter arena (what the C ABI reads; also the all-reduce / Adam buffer) ----
    def _named_in_order(self):
        """The 26 parameters in state_dict() order (cached: nn.Module keeps the Parameter objects across .to() /
        load_state_dict; the cache is checked against the module dict cheaply and rebuilt if a layer was replaced)."""
        pl = self.__dict__.get("_plist")
        if pl is None or pl[0] is not self.imgBranch[0].weight or pl[-1] is not self.soundTriplet[2].bias:
            d = dict(self.named_parameters())
            pl = [d[k] for k, _ in PARAM_SPECS]
            self.__dict__["_plist"] = pl
        return pl

    def _flatten_params(self):
        params = self._named_in_order()
        dev = params[0].device
        flat = torch.empty(N_PARAMS, dtype=torch.float32, device=dev)
        for p, o in zip(params, PARAM_OFFSETS):
            n = p.numel()
            flat[o:o + n].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + n].view(p.shape)
        self._flat = flat

    def _arena_intact(self):
        base = self._flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o and p.dtype == torch.float32
                   for p, o in zip(self._named_in_order(), PARAM_OFFSETS))

    def pack(self, ctx=None):
        """Re-pack this model's kernel-side weight image from the parameters now (after ANY direct edit of them)."""
        return self.hip_weights(ctx, force=True)

    def hip_weights(self, ctx=None, force=False):
        """This model's packed weight image on its device context, bound for the next C-ABI calls; re-packed when the
        parameters changed since the last pack as far as torch's version counters tell (load_state_dict, an external
        optimiser, .to(), in-place ops on the parameters) or when `force`d.  NOT detected: edits through `.data`
        (p.data.mul_(), nn.init on m.weight.data) -- those bump no counter; forward() therefore always forces, and code
        that binds the image itself (trainers, graphs) calls pack() after such edits.  VARTrainer's own Adam keeps the
        image current by itself."""
        flat = self.flat_parameters()
        if ctx is None:
            ctx = Context.get(flat.device.index)
        w = self._weights
        if w is None or w.ctx is not ctx:
            w = self._weights = ctx.new_weights()
        key = (flat.data_ptr(), flat._version, sum(p._version for p in self._named_in_order()))
        if force or w.key != key:
            w.pack(flat, key)
        else:
            w.bind()
        return w

    def flat_parameters(self):
        if not self._arena_intact():
            self._flatten_params()
        return self._flat

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__["_plist"] = None
        self._flatten_params()
        return r

    # ---- forward: PretextNetBase.VAR_forward routing (pretext_base.py:10-41) ----
    def forward(self, image, sound_positive, sound_negative, is_train=False):
        flat = self.flat_parameters()
        if not flat.is_cuda:
            raise VarHipError("VARPretextNet runs on the GPU only: call .to('cuda') (no CPU fallback)")
        dev = flat.device

        def prep(t, name):
            if t is None:
                return None
            if not t.is_cuda:
                raise VarHipError(f"{name} must be a CUDA tensor (no CPU fallback)")
            return t

        image = prep(image, "image")
        sound_positive = prep(sound_positive, "sound_positive")
        sound_negative = prep(sound_negative, "sound_negative")
        if image is not None:
            if image.dtype != torch.uint8:
                image = image.float()
            if image.shape[1] < 3 or tuple(image.shape[2:]) != tuple(self.config.img_dim[1:]):
                raise VarHipError(f"image shape {tuple(image.shape)} does not match img_dim {self.config.img_dim}")
            image = image.contiguous()                   # channels >3 allowed: image[:, :3] (pretext_base.py:22)
        run_pos = sound_positive is not None and (not torch.isinf(sound_positive).all())   # pretext_base.py:29
        pos = sound_positive.float().contiguous() if run_pos else None
        neg = sound_negative.float().contiguous() if sound_negative is not None else None
        for s, nm in ((pos, "sound_positive"), (neg, "sound_negative")):
            if s is not None and tuple(s.shape[1:]) != (1, 100, 40):
                raise VarHipError(f"{nm} shape {tuple(s.shape)} is not (B,1,100,40)")

        image_feat = image_feat_raw = pos_sound_raw = sound_feat_negative = None
        if image is not None or pos is not None or neg is not None:
            params = self._named_in_order()
            need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
            if need_grad:
                outs = _EncoderFn.apply(self, True, image, pos, neg, *params)
            else:                                         # inference: no autograd node, no 26-parameter argument list
                outs = _encoder_forward(self, False, image, pos, neg)[0]
            if image is not None:
                image_feat, image_feat_raw = outs[0], outs[3]
            if pos is not None:
                self.cached_sound = outs[1]
                pos_sound_raw = outs[4]
            if neg is not None:
                sound_feat_negative = outs[2]
        return {'image_feat': image_feat, 'sound_feat_positive': self.cached_sound,
                'sound_feat_negative': sound_feat_negative, 'image_BCE': None, 'sound_BCE': None,
                'image_feat_raw': image_feat_raw, 'pos_sound_raw': pos_sound_raw}
