"""Drop-in for the reference's iTHOR `config.pretextModel` (Envs/ai2thor/config.py:33):
models/pretext/ai2thor_pretext_model.py:VARPretextNet with the same constructor, module names, state_dict
layout (36 tensors) and forward() contract (pretext_base.py:10-41); the arithmetic is libvar_hip.so
(csrc/ithor.hip, var_ithor_* of include/var_hip.h).  GPU only -- no CPU fallback."""
import numpy as np
import torch
import torch.nn as nn

from ._lib import Context, VarHipError, current_stream_handle, new_graph, ptr


class _IthorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, need_grad, image, pos, neg, *params):
        flat = module._flat
        dev = flat.device
        c = Context.get(dev.index)
        ref = image if image is not None else (pos if pos is not None else neg)
        B = ref.shape[0]
        H = module.config.img_dim[1]
        module._ensure_plan(c, B)
        mk = lambda n: torch.empty((B, n), dtype=torch.float32, device=dev)
        image_feat = mk(3) if image is not None else None
        image_raw = mk(1152) if image is not None else None
        pos_feat = mk(3) if pos is not None else None
        pos_raw = mk(1024) if pos is not None else None
        neg_feat = mk(3) if neg is not None else None
        is_u8 = image is not None and image.dtype == torch.uint8
        bstride = 0 if image is None else image.stride(0)
        c.check(c.lib.var_ithor_encoder_fwd(c.handle, current_stream_handle(), ptr(flat), ptr(image), int(is_u8),
                                            bstride, ptr(pos), ptr(neg), B, H, ptr(image_feat), ptr(pos_feat),
                                            ptr(neg_feat), ptr(image_raw), ptr(pos_raw), int(need_grad)),
                "var_ithor_encoder_fwd")
        ctx.module = module
        ctx.keep = (image, pos, neg)
        ctx.gen = c.lib.var_ithor_saved_generation(c.handle) if need_grad else 0
        outs = (image_feat, pos_feat, neg_feat, image_raw, pos_raw)
        ctx.present = [o is not None for o in outs]
        dummy = torch.zeros(0, device=dev)
        res = tuple(o if o is not None else dummy for o in outs)
        ctx.mark_non_differentiable(*res[3:])
        return res

    @staticmethod
    def backward(ctx, g_if, g_pf, g_nf, g_ir, g_pr):
        module = ctx.module
        flat = module._flat
        c = Context.get(flat.device.index)
        if not ctx.gen or c.lib.var_ithor_saved_generation(c.handle) != ctx.gen:
            raise VarHipError("backward of a forward whose activations are gone: the device context keeps ONE saved "
                              "forward (a later forward of this or another model overwrote it, or it ran under "
                              "no_grad) -- run forward and backward back to back")
        # a fresh buffer per backward: autograd may keep the returned views as .grad (AccumulateGrad steals them)
        gflat = torch.empty_like(flat)
        gs = [g.contiguous().float() if (present and g is not None) else None
              for g, present in zip((g_if, g_pf, g_nf), ctx.present[:3])]
        c.check(c.lib.var_ithor_encoder_bwd(c.handle, current_stream_handle(), ptr(flat), ptr(gs[0]), ptr(gs[1]),
                                            ptr(gs[2]), ptr(gflat)), "var_ithor_encoder_bwd")
        grads, o = [], 0
        for p in module._params_in_order():
            grads.append(gflat[o:o + p.numel()].view(p.shape))
            o += p.numel()
        return (None, None, None, None, None, *grads)


class IthorVARPretextNet(nn.Module):
    """iTHOR VAR encoder.  ctor(config) reads config.img_dim / sound_dim / representationDim
    (ai2thor_pretext_model.py:42-58); modules are created in the reference's order (imgBranch, rnn, cnn,
    imgTriplet, soundTriplet) so that a given torch.manual_seed draws the reference's initial weights."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        if tuple(config.sound_dim) != (1, 600, 40) or config.representationDim != 3 or config.img_dim[0] != 3 \
                or config.img_dim[1] != config.img_dim[2] or self._final_side(config.img_dim[1]) != 3:
            raise VarHipError("HIP iTHOR VARPretextNet supports square 3-channel images ending in a 3x3 map "
                              f"(96, 84), sound_dim (1,600,40), representationDim 3; got {config.img_dim} "
                              f"{config.sound_dim} {config.representationDim}")
        self.cached_sound = None
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
        self.imgTriplet = nn.Sequential(nn.Linear(128 * 9, 128), nn.ReLU(), nn.Linear(128, config.representationDim))
        self.soundTriplet = nn.Sequential(nn.Linear(2 * 512, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU(),
                                          nn.Linear(64, config.representationDim))
        self._flat = None
        self._plan = 0
        self._bf16 = 0
        self._flatten_params()

    @staticmethod
    def _final_side(h):
        for _ in range(4):
            h //= 2
        return (h - 1) // 2 + 1

    # ---- flat parameter arena in state_dict order (what the C ABI reads) ----
    def _params_in_order(self):
        return [p for _, p in self.named_parameters()]

    def _flatten_params(self):
        params = self._params_in_order()
        n = sum(p.numel() for p in params)
        flat = torch.empty(n, dtype=torch.float32, device=params[0].device)
        o = 0
        for p in params:
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
            o += p.numel()
        self._flat = flat

    def _arena_intact(self):
        o = self._flat.data_ptr()
        for p in self._params_in_order():
            if p.data_ptr() != o or p.dtype != torch.float32:
                return False
            o += 4 * p.numel()
        return True

    def flat_parameters(self):
        if not self._arena_intact():
            self._flatten_params()
        return self._flat

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        # nn.GRU keeps a list of flattened weights; our arena replaces its storage, which is all it needs
        self._flatten_params()
        return r

    def _ensure_plan(self, c, batch):
        # the workspace belongs to the context, not to this module: ask every time (a no-op when it already fits; another
        # model with a different image size may have re-planned the context in between)
        c.check(c.lib.var_ithor_plan(c.handle, int(batch), int(self.config.img_dim[1])), "var_ithor_plan")
        if c.lib.var_ithor_set_bf16(c.handle, -1) != int(self._bf16):     # the plan is per context, the choice per model
            c.lib.var_ithor_set_bf16(c.handle, int(self._bf16))
        seq = int(getattr(self, "_gru_sequence", True))
        if c.lib.var_ithor_set_gru_sequence(c.handle, -1) != seq:
            c.lib.var_ithor_set_gru_sequence(c.handle, seq)

    def set_gru_sequence(self, on=True):
        """bf16 mode: run each GRU pass as one persistent launch (default) or as one launch per time step (same results).
        The persistent form needs its whole grid resident at once -- switch it off when several processes share one GPU
        (include/var_hip.h: var_ithor_set_gru_sequence; a launch whose waits expire poisons the step with NaN)."""
        self._gru_sequence = bool(on)
        return self

    def gru_status(self, device_index=None):
        """0 if every hand-off of every persistent GRU launch so far completed (blocking read of the status word)."""
        import ctypes
        c = Context.get(self.flat_parameters().device.index if device_index is None else device_index)
        w = ctypes.c_uint(0)
        c.check(c.lib.var_ithor_gru_status(c.handle, ctypes.byref(w)), "var_ithor_gru_status")
        return int(w.value)

    def set_precision(self, name, keep_fp32_activations=False):
        """'fp32' (default, the parity path) or 'bf16': bf16 operands with fp32 accumulation in every product
        (BASELINE config 4's stated precision); parameters, activations, gradients and Adam state stay fp32."""
        if name not in ("fp32", "bf16"):
            raise VarHipError("precision is 'fp32' or 'bf16'")
        # keep_fp32_activations: also store the fp32 copies of the sound CNN's intermediate maps (debug buffers of the
        # layer-wise tests); the bf16 mode itself only needs their bf16 images
        self._bf16 = (2 if keep_fp32_activations else 1) if name == "bf16" else 0
        return self

    def forward(self, image, sound_positive, sound_negative, is_train=False):
        flat = self.flat_parameters()
        if not flat.is_cuda:
            raise VarHipError("IthorVARPretextNet runs on the GPU only: call .to('cuda') (no CPU fallback)")
        if flat.numel() != Context.get(flat.device.index).lib.var_ithor_param_count():
            raise VarHipError("parameter arena does not match var_ithor_param_count()")
        for t, nm in ((image, "image"), (sound_positive, "sound_positive"), (sound_negative, "sound_negative")):
            if t is not None and not t.is_cuda:
                raise VarHipError(f"{nm} must be a CUDA tensor (no CPU fallback)")
        if image is not None:
            if image.dtype != torch.uint8:
                image = image.float()
            if image.shape[1] < 3 or tuple(image.shape[2:]) != tuple(self.config.img_dim[1:]):
                raise VarHipError(f"image shape {tuple(image.shape)} does not match img_dim {self.config.img_dim}")
            image = image.contiguous()
        run_pos = sound_positive is not None and (not torch.isinf(sound_positive).all())   # pretext_base.py:29
        pos = sound_positive.float().contiguous() if run_pos else None
        neg = sound_negative.float().contiguous() if sound_negative is not None else None
        for s, nm in ((pos, "sound_positive"), (neg, "sound_negative")):
            if s is not None and tuple(s.shape[1:]) != (1, 600, 40):
                raise VarHipError(f"{nm} shape {tuple(s.shape)} is not (B,1,600,40)")
        image_feat = image_feat_raw = pos_sound_raw = sound_feat_negative = None
        if image is not None or pos is not None or neg is not None:
            params = self._params_in_order()
            need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
            outs = _IthorFn.apply(self, need_grad, image, pos, neg, *params)
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


class IthorTrainer:
    """The step body of VAR_Pretext.trainRepresentation (VAR/pretext_VAR.py:55-70) for the iTHOR model:
    var_ithor_loss_grad (forward, TripletMarginLoss, backward) + var_adam_step on flat arenas.  Same surface as
    VARTrainer (lr, step, loss, loss_and_grads, allreduce, adam); under torch.distributed each rank takes its shard
    of the global batch, the loss/gradients are scaled by 1/B_global and ONE sum all-reduce (RCCL) carries the
    gradient arena and the loss slot (15.4 MB, BASELINE config 4)."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, margin=1.0,
                 process_group=None):
        self.model = model
        self.lr, self.wd, self.margin, self.betas, self.eps = lr, weight_decay, margin, betas, eps
        flat = model.flat_parameters()
        if not flat.is_cuda:
            raise VarHipError("IthorTrainer needs the model on the GPU (no CPU fallback)")
        self.dev = flat.device
        self.ctx = Context.get(flat.device.index)
        self.n = flat.numel()
        self.gbuf = torch.zeros(self.n + 1, dtype=torch.float32, device=self.dev)   # [gradients | loss]
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self._host_steps = 0
        self._g_step = self._g_lr = None
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)

    @property
    def grads(self):
        return self.gbuf[:self.n]

    @property
    def loss(self):
        return self.gbuf[self.n:]

    @property
    def loss_buf(self):
        return self.loss

    def loss_and_grads(self, image, pos, neg, global_batch=None, feats=False):
        m, c = self.model, self.ctx
        flat = m.flat_parameters()
        for t in (image, pos, neg):
            if t is None or not t.is_cuda:
                raise VarHipError("IthorTrainer needs CUDA tensors (image, pos, neg) -- no CPU fallback")
        B = image.shape[0]
        m._ensure_plan(c, B)
        if image.dtype != torch.uint8:
            image = image.float()
        image, pos, neg = image.contiguous(), pos.float().contiguous(), neg.float().contiguous()
        gb = B * self.world if global_batch is None else global_batch
        out = torch.empty((B, 9), dtype=torch.float32, device=flat.device) if feats else None
        c.check(c.lib.var_ithor_loss_grad(c.handle, current_stream_handle(), ptr(flat), ptr(image),
                                          int(image.dtype == torch.uint8), image.stride(0), ptr(pos), ptr(neg), B,
                                          m.config.img_dim[1], float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                          self.gbuf.data_ptr() + 4 * self.n, ptr(out)), "var_ithor_loss_grad")
        self._keep = (image, pos, neg)
        return self.loss, out

    def use_rccl(self, comm):
        """Route the all-reduce through the C ABI (comm.RcclComm) instead of torch.distributed."""
        self.rccl = comm
        self.world, self.rank = comm.size, comm.rank
        return self

    def allreduce(self):
        if getattr(self, "rccl", None) is not None:
            self.rccl.allreduce(self.gbuf)
        elif self.world > 1:
            torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    @property
    def step_count(self):
        """Optimiser steps APPLIED so far.  The count lives on the device (the Adam kernel advances it itself, and leaves it
        alone when it skips a poisoned step); reading it here synchronises."""
        g = getattr(self, "_g_step", None)
        return int(g.item()) if g is not None else self._host_steps

    @step_count.setter
    def step_count(self, v):
        self._host_steps = int(v)
        if getattr(self, "_g_step", None) is not None:
            self._g_step.fill_(int(v))

    def _device_scalars(self):
        if getattr(self, "_g_step", None) is None:
            self._g_lr = torch.full((1,), float(self.lr), dtype=torch.float32, device=self.dev)
            self._g_step = torch.full((1,), int(self._host_steps), dtype=torch.int32, device=self.dev)

    def _guard(self, c):
        # second guard of the Adam kernels: the loss slot that went through the all-reduce with the gradient.  A rank whose
        # persistent GRU launch timed out contributes NaN to it, so EVERY rank skips that step and the replicas stay identical
        # (the time-out word alone is per rank: the others would have applied the NaN gradient the all-reduce gave them)
        c.check(c.lib.var_ithor_guard_loss(c.handle, self.gbuf.data_ptr() + 4 * self.n), "var_ithor_guard_loss")

    def adam(self):
        c = self.ctx
        flat = self.model.flat_parameters()
        self._device_scalars()
        self._g_lr.fill_(float(self.lr))                       # (the eager loop assigns .lr directly, train_representation)
        self._guard(c)
        c.check(c.lib.var_adam_step_dev(c.handle, current_stream_handle(), ptr(flat), ptr(self.gbuf), ptr(self.exp_avg),
                                        ptr(self.exp_avg_sq), self.n, ptr(self._g_lr), float(self.betas[0]),
                                        float(self.betas[1]), float(self.eps), float(self.wd), ptr(self._g_step)),
                "var_adam_step_dev")

    def step(self, image, pos, neg, global_batch=None):
        """One optimisation step.  bf16 mode with the persistent GRU launches: if a launch of this step timed out (its grid
        was not fully resident -- another process on the GPU) the loss reads NaN and the Adam kernel leaves parameters,
        moments and step count alone (a device-side guard word, csrc/pack_adam.hip); the next step clears the word and
        trains normally; gru_status() keeps a sticky record.  Nothing to check per step on the host."""
        self.loss_and_grads(image, pos, neg, global_batch)
        self.allreduce()
        self.adam()
        return self.loss

    def set_lr(self, lr):
        self.lr = lr
        if getattr(self, "_g_lr", None) is not None:
            self._g_lr.fill_(float(lr))

    def capture_step(self, image, pcm, lens, global_batch=None, _ctx=None, _shared_scalars=False):
        """Capture step_from_pcm over STATIC CUDA tensors (image u8|f32 (B,3,H,H), pcm int16 (2B,n), lens int32 (2B)) into
        a HIP graph and return replay(): the step's ~700 launches (2 x 73 dependent recurrent products and gate kernels
        in each direction of time) then cost one graph launch.  The caller refreshes the static tensors in place between
        replays (e.g. TripletPool.gather(..., out_img=, out_pcm=, out_len=)).  Step count and learning rate live on the
        device (var_adam_step_dev; set_lr updates the latter).  Under data parallelism the all-reduce stays eager
        between two graphs."""
        m, c = self.model, (_ctx or self.ctx)
        flat = m.flat_parameters()
        B = image.shape[0]
        for t in (image, pcm, lens):
            if not (t.is_cuda and t.is_contiguous()):
                raise VarHipError("capture_step needs contiguous CUDA tensors")
        if pcm.dtype != torch.int16 or lens.dtype != torch.int32 or image.dtype not in (torch.uint8, torch.float32):
            raise VarHipError("capture_step: image u8|f32, pcm int16, lens int32")
        frames = m.config.sound_dim[1]
        if _ctx is None:
            self.step_from_pcm(image, pcm, lens, global_batch)      # eager once: workspace plan, front-end tables
        else:                                                       # (capture_epoch_steps: same warm-up on the given context, no optimiser step)
            m._ensure_plan(c, B)
            warm = torch.empty((2 * B, 1, frames, 40), dtype=torch.float32, device=self.dev)
            c.check(c.lib.var_mfcc_psf(c.handle, current_stream_handle(), ptr(pcm), ptr(lens), None, 2 * B, pcm.shape[1], frames,
                                       ptr(warm)), "var_mfcc_psf")
            c.check(c.lib.var_ithor_loss_grad(c.handle, current_stream_handle(), ptr(flat), ptr(image),
                                              int(image.dtype == torch.uint8), image.stride(0), ptr(warm[:B]), ptr(warm[B:]), B,
                                              m.config.img_dim[1], float(self.margin), 1.0, ptr(self.gbuf),
                                              self.gbuf.data_ptr() + 4 * self.n, None), "var_ithor_loss_grad")
            torch.cuda.synchronize()
            del warm
        feats = torch.empty((2 * B, 1, frames, 40), dtype=torch.float32, device=self.dev)
        self._g_feats = feats
        self._device_scalars()
        gb = B * self.world if global_batch is None else global_batch

        def body_grad():
            s = current_stream_handle()
            c.check(c.lib.var_mfcc_psf(c.handle, s, ptr(pcm), ptr(lens), None, 2 * B, pcm.shape[1], frames, ptr(feats)),
                    "var_mfcc_psf")
            c.check(c.lib.var_ithor_loss_grad(c.handle, s, ptr(flat), ptr(image), int(image.dtype == torch.uint8),
                                              image.stride(0), ptr(feats[:B]), ptr(feats[B:]), B, m.config.img_dim[1],
                                              float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                              self.gbuf.data_ptr() + 4 * self.n, None), "var_ithor_loss_grad")

        def body_adam():
            self._guard(c)
            c.check(c.lib.var_adam_step_dev(c.handle, current_stream_handle(), ptr(flat), ptr(self.gbuf), ptr(self.exp_avg),
                                            ptr(self.exp_avg_sq), self.n, ptr(self._g_lr), float(self.betas[0]),
                                            float(self.betas[1]), float(self.eps), float(self.wd), ptr(self._g_step)),
                    "var_adam_step_dev")

        collective = self.world > 1 or getattr(self, "rccl", None) is not None
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        graphs = []
        with torch.cuda.stream(side):
            for bodies in ((body_grad,), (body_adam,)) if collective else ((body_grad, body_adam),):
                g = new_graph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for b in bodies:
                        b()
                graphs.append(g)
        torch.cuda.current_stream().wait_stream(side)
        self._keep = getattr(self, "_keep_all", []) + [(image, pcm, lens, feats, graphs)]
        self._keep_all = self._keep

        def replay():
            graphs[0].replay()
            if collective:
                self.allreduce()
                graphs[1].replay()
            return self.loss
        return replay

    def capture_epoch_steps(self, images, pcm, batch, table, global_batch=None, steps_per_epoch=None, tail_batch=0,
                            tail_global_batch=None):
        """The replayed step over an HBM-resident dataset (TripletPool: u8 images (N,3,H,H), int16 clips (M,n)), the
        iTHOR counterpart of VARTrainer.capture_epoch_steps: `table` is an int32 (rows, 5*batch) tensor of step rows
        [image_index | clip_index (2B) | lens (2B)] (TripletPool.index_table / epoch_index_table).  A replay gathers the row
        into static buffers (three small eager launches: ~30 us against a step of 6-40 ms) and launches the captured step
        (capture_step: python_speech_features front-end, forward, triplet loss, backward, [collective,] Adam).  Ragged epochs
        (the reference's iTHOR default is 500 triplets at batch 128: 128 / 128 / 128 / 116, Envs/ai2thor/config.py:24,41): with
        `tail_batch` every `steps_per_epoch`-th row is the short last batch, packed at the head of its row; a second graph
        over the same workspace runs it.  Returns (replay, load_table)."""
        m = self.model
        B = int(batch)
        rows, row_ints = int(table.shape[0]), int(table.shape[1])
        if row_ints != 5 * B or table.dtype != torch.int32 or table.device != self.dev or not table.is_contiguous():
            raise VarHipError("index table must be a contiguous int32 (rows, 5*batch) tensor on the trainer's device")
        tail_batch = int(tail_batch)
        if tail_batch and (not (0 < tail_batch < B) or not steps_per_epoch or rows % steps_per_epoch):
            raise VarHipError("ragged table: need 0 < tail_batch < batch and rows a multiple of steps_per_epoch")
        if not (images.is_cuda and images.dtype == torch.uint8 and pcm.is_cuda and pcm.dtype == torch.int16
                and images.is_contiguous() and pcm.is_contiguous()):
            raise VarHipError("capture_epoch_steps needs the pool's contiguous CUDA tensors: u8 images, int16 clips")
        sizes = [(B, B * self.world if global_batch is None else int(global_batch))]
        if tail_batch:
            sizes.append((tail_batch, tail_batch * self.world if tail_global_batch is None else int(tail_global_batch)))
        ctxs = [self.ctx, self.ctx]                           # both graphs run on the trainer's workspace (planned for `batch`)
        self._device_scalars()                                 # every graph of this trainer shares ONE device-side step count / learning rate
        plans = []
        for (Bs, gb), cx in zip(sizes, ctxs):
            img = torch.zeros((Bs,) + tuple(images.shape[1:]), dtype=torch.uint8, device=self.dev)
            clp = torch.zeros((2 * Bs, pcm.shape[1]), dtype=torch.int16, device=self.dev)
            lns = torch.zeros(2 * Bs, dtype=torch.int32, device=self.dev)
            idx = torch.zeros(3 * Bs, dtype=torch.int64, device=self.dev)
            plans.append((Bs, img, clp, lns, idx, self.capture_step(img, clp, lns, global_batch=gb, _ctx=cx, _shared_scalars=True)))
        state = {"row": 0, "table": table.clone()}

        def is_tail(r):
            return bool(tail_batch) and r % steps_per_epoch == steps_per_epoch - 1

        def load_table(t):
            assert t.shape == state["table"].shape
            state["table"].copy_(t, non_blocking=True)
            state["row"] = 0

        def replay():
            Bs, img, clp, lns, idx, run = plans[1 if is_tail(state["row"]) else 0]
            r = state["table"][state["row"]]
            torch.index_select(images, 0, r[:Bs], out=img)     # (int32 indices as they are: [image ids | clip ids])
            torch.index_select(pcm, 0, r[Bs:3 * Bs], out=clp)
            lns.copy_(r[3 * Bs:5 * Bs])
            run()
            state["row"] = (state["row"] + 1) % rows
            return self.loss
        return replay, load_table

    def step_from_pcm(self, image, pcm, lens, global_batch=None):
        """The step with the data-loader's audio work folded in (dataset.py:64-89 + Envs/audioLoader.py:158-161,
        241-252): pcm int16 (2B, n) = [pos | neg] clips resident in HBM (e.g. TripletPool.gather), lens (2B) valid
        samples (0 = the "empty" class => all-zero features); python_speech_features MFCC on the GPU, then step."""
        from .ops import mfcc_psf
        B = image.shape[0]
        feats = mfcc_psf(pcm, lens, out_frames=self.model.config.sound_dim[1])
        return self.step(image, feats[:B], feats[B:], global_batch)


def project_representation(model, batches):
    """pretext.py:147-203 (project2representation_with_ground_truth) without the plotting: run the frozen encoder
    over `batches()` = (image, sound_positive, sound_negative, gt) and return (image_feat (N,3), sound_feat (N,3),
    gt (N,)) numpy arrays for the scatter / t-SNE check.  Works for both VARPretextNet models."""
    was_training = model.training
    model.eval()
    img, snd, gts = [], [], []
    with torch.no_grad():
        for image, sp, _sn, gt in batches():
            d = model(image, sp, None)
            img.append(d['image_feat'].cpu().numpy().copy())
            snd.append(d['sound_feat_positive'].cpu().numpy().copy())
            gts.append(np.asarray(gt.cpu() if torch.is_tensor(gt) else gt).reshape(-1))
    model.train(was_training)
    return np.concatenate(img), np.concatenate(snd), np.concatenate(gts)
