"""TEST INFRASTRUCTURE.  A CPU stand-in for the device context that VARTrainer talks to (`_ctx=`): the same C-ABI
entry names (include/var_hip.h) bound to the ORACLE (oracle/torch_oracle.py, oracle/mfcc_np.py) on host memory, and
"graph capture" as plain closures.  It lets the CPU suite drive VARTrainer's own host logic -- shard scaling,
buffer slots, index-row layouts, rank offsets, order of collectives and replays -- with world_size 2 on gloo
(tests/test_dp_gloo.py), and the ragged-epoch bookkeeping (tests/test_trainer_host.py).  Never imported by the product."""
import ctypes

import numpy as np
import torch

from oracle import mfcc_np
from oracle.torch_oracle import KukaNetCPU, inbatch_contrastive_loss
from oracle.var_oracle import PARAM_SPECS, N_PARAMS


def _arr(p, n, dtype=np.float32):
    """numpy view of n elements of host memory at address p (the tensors' data_ptr())."""
    if not p:
        return None
    ct = {np.float32: ctypes.c_float, np.int32: ctypes.c_int32, np.int16: ctypes.c_int16, np.uint8: ctypes.c_uint8}[dtype]
    return np.ctypeslib.as_array((ct * int(n)).from_address(int(p)))


class _Weights:
    def __init__(self, ctx):
        self.ctx, self.key, self.binds, self.packs = ctx, None, 0, 0

    def bind(self):
        self.binds += 1

    def pack(self, flat, key=None):
        self.packs += 1
        self.key = key


class _Lib:
    """The entries VARTrainer uses, computed by the oracle.  Signatures as in include/var_hip.h."""

    def __init__(self):
        self.calls = []
        self._saved = None

    # -- helpers
    def _net(self, flat_ptr):
        flat = torch.from_numpy(_arr(flat_ptr, N_PARAMS).copy())
        net = KukaNetCPU()
        sd, o = {}, 0
        for k, s in PARAM_SPECS:
            n = int(np.prod(s))
            sd[k] = flat[o:o + n].view(s)
            o += n
        net.load_state_dict(sd)
        return net

    def _images(self, image, is_u8, bstride, idx_ptr, B, H):
        rows = np.arange(B) if not idx_ptr else _arr(idx_ptr, B, np.int32)
        n_rows = int(rows.max()) + 1
        a = _arr(image, n_rows * bstride, np.uint8 if is_u8 else np.float32).reshape(n_rows, bstride)[rows]
        img = torch.from_numpy(np.ascontiguousarray(a[:, :3 * H * H])).view(B, 3, H, H)
        return img.float() / 255. if is_u8 else img.float()

    def _grad_arena(self, net):
        return torch.cat([dict(net.named_parameters())[k].grad.reshape(-1) for k, _ in PARAM_SPECS])

    def _loss_grad(self, flat, img, pos, neg, margin, inv_count, grads, loss):
        net = self._net(flat)
        a, p, n = net(img, pos, neg)
        ls = torch.nn.TripletMarginLoss(margin=margin, p=2, reduction="sum")(a, p, n) * inv_count
        ls.backward()
        _arr(grads, N_PARAMS)[:] = self._grad_arena(net).numpy()
        _arr(loss, 1)[0] = float(ls.detach())
        return 0

    def _mfcc(self, pcm, pcm_stride, clip_idx, lens, n, frames):
        rows = np.arange(n) if not clip_idx else _arr(clip_idx, n, np.int32)
        ln = _arr(lens, n, np.int32)
        src = _arr(pcm, (int(rows.max()) + 1) * pcm_stride, np.int16).reshape(-1, pcm_stride)
        out = np.zeros((n, 1, frames, 40), np.float32)
        for i in range(n):
            if ln[i] > 0:
                f = mfcc_np.mfcc_torchaudio(src[rows[i], :ln[i]], dtype=np.float32)
                out[i] = mfcc_np.process_sound_feat(f, (1, frames, 40))
        return out

    # -- entries
    def var_arm_loss_grad(self, h, s, flat, image, is_u8, bstride, pos, neg, B, H, margin, inv, grads, loss, feats):
        self.calls.append(("loss_grad", B, inv))
        img = self._images(image, is_u8, bstride, None, B, H)
        mf = lambda p: torch.from_numpy(_arr(p, B * 4000).copy()).view(B, 1, 100, 40)  # noqa: E731
        return self._loss_grad(flat, img, mf(pos), mf(neg), margin, inv, grads, loss)

    def var_arm_loss_grad_gather(self, h, s, flat, image, is_u8, bstride, idx, pos, neg, B, H, margin, inv, grads, loss, feats):
        self.calls.append(("loss_grad_gather", B, inv))
        img = self._images(image, is_u8, bstride, idx, B, H)
        mf = lambda p: torch.from_numpy(_arr(p, B * 4000).copy()).view(B, 1, 100, 40)  # noqa: E731
        return self._loss_grad(flat, img, mf(pos), mf(neg), margin, inv, grads, loss)

    def var_arm_loss_grad_pcm(self, h, s, flat, image, is_u8, bstride, idx, pcm, pcm_stride, clip_idx, lens, B, H, margin,
                              inv, grads, loss, feats):
        self.calls.append(("loss_grad_pcm", B, inv))
        img = self._images(image, is_u8, bstride, idx, B, H)
        f = torch.from_numpy(self._mfcc(pcm, pcm_stride, clip_idx, lens, 2 * B, 100))
        return self._loss_grad(flat, img, f[:B], f[B:], margin, inv, grads, loss)

    def var_mfcc(self, h, s, pcm, lens, clip_idx, n, pcm_stride, frames, out):
        self.calls.append(("mfcc", n))
        _arr(out, n * frames * 40)[:] = self._mfcc(pcm, pcm_stride, clip_idx, lens, n, frames).reshape(-1)
        return 0

    def _adam(self, flat, g, m, v, n, lr, b1, b2, eps, wd, step):
        p = torch.from_numpy(_arr(flat, n))
        gg = torch.from_numpy(_arr(g, n)) + wd * p
        mm, vv = torch.from_numpy(_arr(m, n)), torch.from_numpy(_arr(v, n))
        mm.lerp_(gg, 1 - b1)
        vv.mul_(b2).addcmul_(gg, gg, value=1 - b2)
        denom = (vv.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
        p.addcdiv_(mm, denom, value=-lr / (1 - b1 ** step))

    def var_adam_step(self, h, s, flat, g, m, v, n, lr, b1, b2, eps, wd, step):
        self.calls.append(("adam", step))
        self._adam(flat, g, m, v, n, lr, b1, b2, eps, wd, step)
        return 0

    def var_adam_step_graph(self, h, s, flat, g, m, v, n, lr_dev, b1, b2, eps, wd, step_dev, table, row_ints, rows, cursor,
                            idx_row, ahead):
        st = _arr(step_dev, 1, np.int32)
        st[0] += 1
        self.calls.append(("adam_graph", int(st[0])))
        self._adam(flat, g, m, v, n, float(_arr(lr_dev, 1)[0]), b1, b2, eps, wd, int(st[0]))
        if table:
            tab = _arr(table, rows * row_ints, np.int32).reshape(rows, row_ints)
            cur = _arr(cursor, 1, np.int32)
            nxt = (int(cur[0]) + 1) % rows
            dst = _arr(idx_row, (2 if ahead else 1) * row_ints, np.int32)
            dst[:row_ints] = tab[nxt]
            if ahead:
                dst[row_ints:] = tab[(nxt + 1) % rows]
            cur[0] = nxt
        return 0

    def var_arm_encoder_fwd(self, h, s, flat, image, is_u8, bstride, pos, neg, B, H, f_img, f_pos, f_neg, raw_i, raw_p, save):
        self.calls.append(("encoder_fwd", B))
        net = self._net(flat)
        img = self._images(image, is_u8, bstride, None, B, H)
        mf = lambda p: torch.from_numpy(_arr(p, B * 4000).copy()).view(B, 1, 100, 40)  # noqa: E731
        a, p, n = net(img, mf(pos), mf(neg))
        for dst, t in ((f_img, a), (f_pos, p), (f_neg, n)):
            _arr(dst, 3 * B)[:] = t.detach().numpy().reshape(-1)
        self._saved = (net, a, p, n) if save else None
        return 0

    def var_arm_encoder_bwd(self, h, s, flat, ga, gp, gn, grads):
        self.calls.append(("encoder_bwd",))
        net, a, p, n = self._saved
        B = a.shape[0]
        g = lambda q: torch.from_numpy(_arr(q, 3 * B).copy()).view(B, 3)  # noqa: E731
        torch.autograd.backward([a, p, n], [g(ga), g(gp), g(gn)])
        _arr(grads, N_PARAMS)[:] = self._grad_arena(net).numpy()
        return 0

    def var_inbatch_loss_fwd_bwd(self, h, s, anchor, cand, target, B, M, tau, inv, scratch, loss, ga, gc):
        self.calls.append(("inbatch", B, M, inv))
        a = torch.from_numpy(_arr(anchor, 3 * B).copy()).view(B, 3).requires_grad_()
        c = torch.from_numpy(_arr(cand, 3 * M).copy()).view(M, 3).requires_grad_()
        t = torch.from_numpy(_arr(target, B, np.int32).copy()).long()
        ls = inbatch_contrastive_loss(a, c, t, tau=tau, inv_count=inv)
        ls.backward()
        _arr(loss, 1)[0] = float(ls)
        _arr(ga, 3 * B)[:] = a.grad.numpy().reshape(-1)
        _arr(gc, 3 * M)[:] = c.grad.numpy().reshape(-1)
        return 0


class OracleContext:
    """What VARTrainer needs of a device context (`_lib.Context`), on the CPU."""

    def __init__(self):
        self.lib = _Lib()
        self.handle = None
        self.plans = []

    def check(self, rc, what):
        assert rc == 0, what

    def join_timeouts(self):
        return 0                                            # (no device, no device-side hand-over)

    def ensure_plan(self, batch, hw):
        self.plans.append((batch, hw))

    def stream(self):
        return None

    def new_weights(self):
        return _Weights(self)

    def capture(self, groups):
        def runner(bodies):
            return lambda: [b() for b in bodies]
        return [runner(list(bodies)) for bodies in groups]
