"""ORACLE loader (test infrastructure only).

ctypes front-end of oracle/libvar_oracle.so (var_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# state_dict() registration order of the reference Kuka VARPretextNet
# (models/pretext/arm_pretext_model.py:39-56) with the PyTorch shapes.
PARAM_SPECS = [
    ("imgBranch.0.weight", (32, 3, 3, 3)), ("imgBranch.0.bias", (32,)),
    ("imgBranch.2.weight", (32, 32, 3, 3)), ("imgBranch.2.bias", (32,)),
    ("imgBranch.4.weight", (64, 32, 3, 3)), ("imgBranch.4.bias", (64,)),
    ("imgBranch.6.weight", (64, 64, 3, 3)), ("imgBranch.6.bias", (64,)),
    ("imgBranch.8.weight", (64, 64, 3, 3)), ("imgBranch.8.bias", (64,)),
    ("soundCNN.0.weight", (32, 1, 5, 40)), ("soundCNN.0.bias", (32,)),
    ("soundCNN.2.weight", (32, 32, 3, 1)), ("soundCNN.2.bias", (32,)),
    ("soundCNN.4.weight", (32, 32, 3, 1)), ("soundCNN.4.bias", (32,)),
    ("soundCNN.6.weight", (32, 32, 3, 1)), ("soundCNN.6.bias", (32,)),
    ("imgTriplet.0.weight", (128, 576)), ("imgTriplet.0.bias", (128,)),
    ("imgTriplet.2.weight", (3, 128)), ("imgTriplet.2.bias", (3,)),
    ("soundTriplet.0.weight", (128, 160)), ("soundTriplet.0.bias", (128,)),
    ("soundTriplet.2.weight", (3, 128)), ("soundTriplet.2.bias", (3,)),
]
N_PARAMS = sum(int(np.prod(s)) for _, s in PARAM_SPECS)


def build(force=False):
    so = os.path.join(_HERE, "libvar_oracle.so")
    src = os.path.join(_HERE, "var_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvar_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_kuka_loss_grad.restype = ctypes.c_float
        _LIB.orc_triplet.restype = ctypes.c_float
        assert _LIB.orc_param_count() == N_PARAMS
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def flatten_params(sd):
    """dict name->array  ->  flat f32 arena in PARAM_SPECS order."""
    return np.concatenate([np.asarray(sd[k], dtype=np.float32).reshape(-1) for k, _ in PARAM_SPECS])


def unflatten_params(flat):
    out, o = {}, 0
    for k, s in PARAM_SPECS:
        n = int(np.prod(s))
        out[k] = flat[o:o + n].reshape(s)
        o += n
    return out


def _split_image(image):
    image = np.ascontiguousarray(image)
    if image.dtype == np.uint8:
        return image, None
    return None, _f32(image)


def forward(params, image, pos, neg):
    """Encoder forward.  image: u8 or f32 (B,3,H,H) or None; pos/neg f32 (B,1,100,40) or None."""
    P = _f32(params)
    ref = image if image is not None else (pos if pos is not None else neg)
    B = ref.shape[0]
    H = image.shape[-1] if image is not None else 84
    if image is None:
        u8, f32 = None, np.zeros((B, 3, H, H), np.float32)
    else:
        u8, f32 = _split_image(image)
    pos, neg = _f32(pos), _f32(neg)
    out = dict(image_feat=np.zeros((B, 3), np.float32), sound_feat_positive=np.zeros((B, 3), np.float32),
               sound_feat_negative=np.zeros((B, 3), np.float32), image_feat_raw=np.zeros((B, 576), np.float32),
               pos_sound_raw=np.zeros((B, 160), np.float32))
    lib().orc_kuka_forward(_p(P), B, H, _p(u8), _p(f32), _p(pos), _p(neg), _p(out['image_feat']),
                           _p(out['sound_feat_positive']), _p(out['sound_feat_negative']),
                           _p(out['image_feat_raw']), _p(out['pos_sound_raw']))
    if image is None:
        out['image_feat'] = out['image_feat_raw'] = None
    if pos is None:
        out['sound_feat_positive'] = out['pos_sound_raw'] = None
    if neg is None:
        out['sound_feat_negative'] = None
    return out


def loss_grad(params, image, pos, neg, margin=1.0):
    """fwd + TripletMarginLoss + backward.  Returns (loss, flat grads, (a, p, n) embeddings)."""
    P = _f32(params)
    B, H = image.shape[0], image.shape[-1]
    u8, f32 = _split_image(image)
    pos, neg = _f32(pos), _f32(neg)
    G = np.zeros(N_PARAMS, np.float32)
    a, p, n = (np.zeros((B, 3), np.float32) for _ in range(3))
    loss = lib().orc_kuka_loss_grad(_p(P), _p(G), B, H, _p(u8), _p(f32), _p(pos), _p(neg),
                                    ctypes.c_float(margin), _p(a), _p(p), _p(n))
    return float(loss), G, (a, p, n)


def triplet(a, p, n, margin=1.0, want_grads=True):
    a, p, n = _f32(a), _f32(p), _f32(n)
    B = a.shape[0]
    ga, gp, gn = (np.zeros((B, 3), np.float32) for _ in range(3))
    loss = lib().orc_triplet(_p(a), _p(p), _p(n), B, ctypes.c_float(margin),
                             _p(ga) if want_grads else None, _p(gp), _p(gn))
    return float(loss), ga, gp, gn


def adam(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-6):
    """In-place torch.optim.Adam step (1-based step index) on flat f32 arrays."""
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    lib().orc_adam(_p(p), _p(g), _p(m), _p(v), ctypes.c_long(p.size), ctypes.c_float(lr), ctypes.c_float(b1),
                   ctypes.c_float(b2), ctypes.c_float(eps), ctypes.c_float(wd), int(step))


def conv3x3s2_fwd(x, w, b):
    x, w, b = _f32(x), _f32(w), _f32(b)
    B, Ci, H, _ = x.shape
    Co = w.shape[0]
    Ho = (H - 1) // 2 + 1
    y = np.zeros((B, Co, Ho, Ho), np.float32)
    lib().orc_conv3x3s2_fwd(_p(x), _p(w), _p(b), _p(y), B, Ci, H, Co)
    return y


def conv3x3s2_bwd(x, w, y, dy, need_dx=True):
    """dy: grad wrt post-ReLU y (masked internally by y>0).  Returns dx, dw, db."""
    x, w, y = _f32(x), _f32(w), _f32(y)
    dy = _f32(dy).copy()
    B, Ci, H, _ = x.shape
    Co = w.shape[0]
    dx = np.zeros_like(x) if need_dx else None
    dw = np.zeros_like(w)
    db = np.zeros(Co, np.float32)
    lib().orc_conv3x3s2_bwd(_p(x), _p(w), _p(y), _p(dy), _p(dx), _p(dw), _p(db), B, Ci, H, Co)
    return dx, dw, db


def multistep_lr(base_lr, milestones, gamma, epoch):
    """utils.py:42-46 get_scheduler -> MultiStepLR: lr in effect during `epoch` (0-based)."""
    lr = base_lr
    for ms in sorted(milestones):
        if epoch >= ms:
            lr = lr * gamma
    return lr
