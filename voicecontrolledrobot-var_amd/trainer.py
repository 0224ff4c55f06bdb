"""Counterpart of VAR_Pretext.trainRepresentation (VAR/pretext_VAR.py:16-95): the step body
zero_grad -> forward -> TripletMarginLoss -> backward -> Adam.step as fused HIP launches on flat
arenas, per-epoch MultiStepLR (utils.py:42-46), periodic legacy-format checkpoints and
progress.csv.  Data parallel: one process per GPU, one RCCL all-reduce of the flat gradient
arena (+ the loss scalar riding in its last slot) per step."""
import csv
import os

import torch

from ._lib import Context, VarHipError, ptr
from .layout import N_PARAMS
from .ops import mfcc as mfcc_op  # noqa: F401


def multistep_lr(base_lr, milestones, gamma, epoch):
    """lr in effect during 0-based `epoch` under MultiStepLR(milestones, gamma) stepped once per epoch."""
    lr = base_lr
    for ms in sorted(milestones):
        if epoch >= ms:
            lr *= gamma
    return lr


def _dist_on(pg):
    return pg is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())


class VARTrainer:
    """The step body of VAR/pretext_VAR.py:55-70 on the HIP library.

    `_ctx` is the device context the launches go through; by default the HIP context of the model's GPU
    (`Context.get`), and there is no other one in the product: a model on the CPU raises.  The parameter exists so
    that the host logic of this class -- shard scaling, buffer slots, rank offsets, the order of collectives and
    replays -- can be driven by the CPU test-suite (tests/_oracle_ctx.py binds the same C-ABI names to the oracle)
    with world_size 2 on gloo."""

    def __init__(self, model, lr=1e-4, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, margin=1.0,
                 process_group=None, _ctx=None):
        flat = model.flat_parameters()
        if _ctx is None:
            if not flat.is_cuda:
                raise VarHipError("VARTrainer needs the model on a GPU (no CPU fallback)")
            _ctx = Context.get(flat.device.index)
        self.model = model
        self.dev = flat.device
        self.ctx = _ctx
        self.lr, self.wd, self.betas, self.eps, self.margin = lr, weight_decay, betas, eps, margin
        self.hw = model.config.img_dim[1]
        # gradient arena with one extra slot for the loss so that ONE all-reduce carries both
        self.gbuf = torch.zeros(N_PARAMS + 1, dtype=torch.float32, device=self.dev)
        self.exp_avg = torch.zeros(N_PARAMS, dtype=torch.float32, device=self.dev)
        self.exp_avg_sq = torch.zeros(N_PARAMS, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self.pg = process_group
        self.world, self.rank = 1, 0
        self.rccl = None
        # VAR_FORCE_ALLREDUCE=1: run the data-parallel code path (RCCL all-reduce between two graphs) even with
        # one rank, to rehearse the multi-GPU step on a single-GPU box
        self.force_collective = os.environ.get("VAR_FORCE_ALLREDUCE") == "1"
        if _dist_on(process_group):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
            if self.world > 1:
                # every replica starts from rank 0's parameters (ranks built from different seeds / checkpoints would
                # otherwise drift apart silently: Adam is applied redundantly, never re-synchronised)
                src = torch.distributed.get_global_rank(process_group, 0) if process_group is not None else 0
                torch.distributed.broadcast(flat, src=src, group=process_group)
        self.pack()

    def pack(self):
        """Refresh this model's packed weight image from the parameter arena (after loading a checkpoint or any direct
        edit of the parameters; every optimiser step of this trainer keeps it current by itself)."""
        self.weights = self.model.hip_weights(self.ctx, force=True)

    def _bind(self):
        """Bind this model's packed weight image for the next C-ABI calls -- and re-pack it first if the parameters
        were changed behind the trainer's back (load_state_dict, an external optimiser: torch's version counters tell).
        The trainer's own Adam keeps the image current by itself."""
        self.weights = self.model.hip_weights(self.ctx)

    @property
    def grads(self):
        return self.gbuf[:N_PARAMS]

    @property
    def loss(self):
        """Device scalar: the (global) mean triplet loss of the last step; reading it syncs."""
        return self.gbuf[N_PARAMS:]

    def _loss_ptr(self):
        return self.gbuf.data_ptr() + 4 * N_PARAMS

    def sync_global_batch(self, local_batch):
        """Sum of the ranks' local batch sizes (one small all-reduce, synchronises): the `global_batch` to pass when the
        shards of a step may differ in size -- the default `B_local * world` is only right for equal shards."""
        if self.rccl is not None and self.world > 1:
            t = torch.tensor([float(local_batch)], dtype=torch.float32, device=self.dev)
            self.rccl.allreduce(t)
            return int(round(float(t.item())))
        if self.world > 1:
            t = torch.tensor([int(local_batch)], dtype=torch.int64, device=self.dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            return int(t.item())
        return int(local_batch)

    def _gb(self, B, global_batch):
        return B * self.world if global_batch is None else int(global_batch)

    def loss_and_grads(self, image, pos, neg, global_batch=None):
        """fwd + loss + bwd into the gradient arena (no optimiser step, no collective)."""
        flat = self.model.flat_parameters()
        B = image.shape[0]
        c = self.ctx
        c.ensure_plan(B, self.hw)
        self._bind()
        c.check(c.lib.var_arm_loss_grad(c.handle, c.stream(), ptr(flat), ptr(image),
                                        int(image.dtype == torch.uint8), image.stride(0), ptr(pos), ptr(neg),
                                        B, self.hw, float(self.margin), 1.0 / self._gb(B, global_batch), ptr(self.gbuf),
                                        self._loss_ptr(), None), "var_arm_loss_grad")

    def use_rccl(self, comm):
        """Route the gradient all-reduce through the C ABI (comm.RcclComm, var_allreduce_grads) instead of
        torch.distributed; `comm.size` / `comm.rank` become the world size / rank of the loss and gradient scaling."""
        self.rccl = comm
        self.world, self.rank = comm.size, comm.rank
        return self

    def _collective(self):
        return self.rccl is not None or self.world > 1 or (self.force_collective and torch.distributed.is_initialized())

    def allreduce(self, async_op=False):
        """The ONE exchange of the data-parallel step: in-place sum of [gradients | loss] over the ranks."""
        if self.rccl is not None:
            self.rccl.allreduce(self.gbuf)
        elif self.world > 1 or (self.force_collective and torch.distributed.is_initialized()):
            return torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                async_op=async_op)
        return None

    def adam(self):
        self.step_count += 1
        flat = self.model.flat_parameters()
        c = self.ctx
        self._bind()
        c.check(c.lib.var_adam_step(c.handle, c.stream(), ptr(flat), ptr(self.gbuf),
                                    ptr(self.exp_avg), ptr(self.exp_avg_sq), N_PARAMS, float(self.lr),
                                    float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                                    int(self.step_count)), "var_adam_step")

    # ---- HIP-graph replay of the whole step (launch-bound otherwise: ~20 kernels + stream fork/joins) ----
    def _device_scalars(self, B):
        dev = self.dev
        self._g_lr = torch.full((1,), float(self.lr), dtype=torch.float32, device=dev)
        self._g_step = torch.full((1,), int(self.step_count), dtype=torch.int32, device=dev)

    def sync_device_scalars(self):
        """After eager steps taken between a capture and its replays: the replayed optimiser reads its step count and
        learning rate from device memory, where the capture put them."""
        if getattr(self, "_g_step", None) is not None:
            self._g_step.fill_(int(self.step_count))
            self._g_lr.fill_(float(self.lr))

    def _body_grad_pcm(self, images, pcm, idx, Bs, gb):
        """Closure enqueueing gather + MFCC + fwd + loss + bwd for `Bs` samples whose indices sit PACKED at the head of
        `idx`: [image_index (Bs) | clip_index (2 Bs) | lens (2 Bs)] (a full row is exactly that with Bs = batch; the short
        last batch of an epoch uses the same layout with Bs < batch and the rest of the row unused)."""
        c, flat = self.ctx, self.model.flat_parameters()
        img_idx, clip_idx, lens = idx[:Bs], idx[Bs:3 * Bs], idx[3 * Bs:5 * Bs]

        def body():
            self._bind()
            c.check(c.lib.var_arm_loss_grad_pcm(c.handle, c.stream(), ptr(flat), ptr(images),
                                                int(images.dtype == torch.uint8), images.stride(0), ptr(img_idx),
                                                ptr(pcm), pcm.stride(0), ptr(clip_idx), ptr(lens), Bs, self.hw,
                                                float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                                self._loss_ptr(), None), "var_arm_loss_grad_pcm")
        return body

    def _body_adam(self, table=None, rows=0, row_ints=0, ahead=0):
        c, flat = self.ctx, self.model.flat_parameters()

        def body():
            self._bind()
            c.check(c.lib.var_adam_step_graph(c.handle, c.stream(), ptr(flat), ptr(self.gbuf),
                                              ptr(self.exp_avg), ptr(self.exp_avg_sq), N_PARAMS, ptr(self._g_lr),
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              float(self.wd), ptr(self._g_step),
                                              ptr(table) if table is not None else None, row_ints, rows,
                                              ptr(self._g_cursor) if table is not None else None,
                                              ptr(self._g_idx) if table is not None else None, int(ahead)),
                    "var_adam_step_graph")
        return body

    def capture_dataset_step(self, images, pcm, batch, global_batch=None):
        """Capture step_from_dataset(images, idx, pcm, clip_idx, lens) once; returns replay(idx_row) where
        idx_row is an int32 tensor of 5*batch entries [image_index | clip_index (2B) | lens (2B)].
        Step count and learning rate live on the device (var_adam_step_dev); set_lr() updates the latter."""
        B = batch
        self._g_idx = torch.zeros(5 * B, dtype=torch.int32, device=self.dev)
        self._device_scalars(B)
        self.ctx.ensure_plan(B, self.hw)
        grad = self._body_grad_pcm(images, pcm, self._g_idx, B, self._gb(B, global_batch))
        adam = self._body_adam()
        if self._collective():                           # the all-reduce stays eager between two graphs
            g_grad, g_adam = self.ctx.capture([[grad], [adam]])
        else:
            (g_grad,), g_adam = self.ctx.capture([[grad, adam]]), None

        def replay(idx_row):
            self._bind()
            self._g_idx.copy_(idx_row, non_blocking=True)
            g_grad()
            if g_adam is not None:
                self.allreduce()
                g_adam()
            self.step_count += 1
            return self.loss
        return replay

    def capture_epoch_steps(self, images, pcm, batch, table, global_batch=None, steps_per_epoch=None, tail_batch=0,
                            tail_global_batch=None):
        """The replayed step with the data-loader cursor on the device too: `table` is an int32 tensor (rows, 5*batch)
        of step rows [image_index | clip_index (2B) | lens (2B)] (one or more shuffled epochs,
        TripletPool.index_table).  Returns (replay, load_table): replay() launches the captured step -- no host-side
        copy, the step itself fetches the next row (var_adam_step_graph); load_table(t) installs a new table of the
        same shape (next epochs) and rewinds the cursor.  After `rows` replays without a new table the walk starts over.

        Ragged epochs (the reference's DataLoader has drop_last=False, VAR/pretext_VAR.py:24): with `tail_batch` > 0
        every `steps_per_epoch`-th row is the short last batch of its epoch, `tail_batch` samples packed at the head
        of the row (TripletPool.epoch_index_table(drop_last=False)); a second graph captured for that size runs it,
        with the loss averaged over `tail_batch` samples as TripletMarginLoss does.  rows % steps_per_epoch == 0."""
        B = batch
        rows, row_ints = int(table.shape[0]), int(table.shape[1])
        if row_ints != 5 * B or table.dtype != torch.int32 or table.device != self.dev or not table.is_contiguous():
            raise VarHipError("index table must be a contiguous int32 (rows, 5*batch) tensor on the trainer's device")
        tail_batch = int(tail_batch)
        if tail_batch:
            if not (0 < tail_batch < B) or not steps_per_epoch or rows % steps_per_epoch:
                raise VarHipError("ragged table: need 0 < tail_batch < batch and rows a multiple of steps_per_epoch")
        self._g_table = torch.empty_like(table)
        self._g_cursor = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._device_scalars(B)
        c = self.ctx
        c.ensure_plan(B, self.hw)
        dp = self._collective()
        ahead = 1 if dp else 0
        # data parallel: index_row holds [row k | row k+1]; the audio front-end runs one step ahead (second copy)
        self._g_idx = torch.zeros((2 if dp else 1) * row_ints, dtype=torch.int32, device=self.dev)
        adam = self._body_adam(self._g_table, rows, row_ints, ahead)
        sizes = [(B, self._gb(B, global_batch))]
        if tail_batch:
            sizes.append((tail_batch, self._gb(tail_batch, tail_global_batch)))
        state = {"row": 0}

        def is_tail(row):
            return bool(tail_batch) and row % steps_per_epoch == steps_per_epoch - 1

        if not dp:
            graphs = [c.capture([[self._body_grad_pcm(images, pcm, self._g_idx, Bs, gb), adam]])[0] for Bs, gb in sizes]

            def load_table(t):
                assert t.shape == self._g_table.shape
                self._g_table.copy_(t, non_blocking=True)
                self._g_idx.copy_(t[0], non_blocking=True)
                self._g_cursor.zero_()
                state["row"] = 0

            def replay():
                self._bind()
                graphs[1 if is_tail(state["row"]) else 0]()
                state["row"] = (state["row"] + 1) % rows
                self.step_count += 1
                return self.loss
            load_table(table)
            return replay, load_table

        # ---- data parallel.  The gradient all-reduce has nothing to overlap with inside the step -- every gradient is
        # complete only at the end of the backward -- but the audio front-end of the NEXT step does not depend on the
        # weights: per step
        #     [gather + fwd + loss + bwd, MFCC features of this step precomputed]
        #     -> all_reduce || MFCC of the next step
        #     -> [Adam + re-pack + row fetch]
        # so up to an MFCC kernel's worth (45 us) of collective latency is hidden (body_step below: one graph).
        flat = self.model.flat_parameters()
        self._g_mfcc = torch.zeros(2 * B, 1, 100, 40, dtype=torch.float32, device=self.dev)
        cur, nxt = self._g_idx[:row_ints], self._g_idx[row_ints:]

        def body_grad(Bs, gb):
            img_idx = cur[:Bs]

            def body():
                self._bind()
                c.check(c.lib.var_arm_loss_grad_gather(c.handle, c.stream(), ptr(flat), ptr(images),
                                                       int(images.dtype == torch.uint8), images.stride(0), ptr(img_idx),
                                                       ptr(self._g_mfcc), ptr(self._g_mfcc[Bs:]), Bs, self.hw,
                                                       float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                                       self._loss_ptr(), None), "var_arm_loss_grad_gather")
            return body

        def body_front(Bs):
            clip_idx, lens = nxt[Bs:3 * Bs], nxt[3 * Bs:5 * Bs]

            def body():
                c.check(c.lib.var_mfcc(c.handle, c.stream(), ptr(pcm), ptr(lens), ptr(clip_idx), 2 * Bs,
                                       pcm.stride(0), 100, ptr(self._g_mfcc)), "var_mfcc")
            return body

        body_front(B)()                                    # warm-up outside capture
        g_front = [c.capture([[body_front(Bs)]])[0] for Bs, _ in sizes]      # (features of row 0, load_table)

        # ONE graph per step (round 3): [gather + fwd + loss + bwd] -> { the collective on the capturing stream || MFCC of
        # the NEXT step on a forked stream } -> [Adam + re-pack + row fetch].  RCCL collectives are capturable, so the
        # replay is one graph launch -- round 2 ran three graphs, an eager collective and a host-side work.wait() per step
        # (1.08x the single-process step in the one-rank rehearsal).  One graph per (size of this batch, size of the next)
        # pair that the table can produce.
        def body_step(Bs, gb, Bn):
            grad, front = body_grad(Bs, gb), body_front(Bn)

            def body():
                grad()
                if self.dev.type == "cuda":
                    main = torch.cuda.current_stream(self.dev)
                    aux = self._aux_stream()
                    aux.wait_stream(main)                  # fork (a graph edge under capture)
                    with torch.cuda.stream(aux):
                        front()                            # next step's features: independent of the weights
                    self.allreduce()                       # in flight beside the front-end
                    main.wait_stream(aux)                  # join
                else:                                      # (host rehearsal of the schedule, tests/_oracle_ctx.py)
                    self.allreduce()
                    front()
                adam()
            return body

        def pairs():
            out = {(0, 0)}
            if tail_batch:
                out |= {(0, 1), (1, 0)} if steps_per_epoch > 1 else {(1, 1)}
            return sorted(out)
        # capturable: RCCL (the C ABI's communicator, or torch.distributed's nccl backend); a gloo group on CUDA tensors
        # stages through the host and cannot be recorded -- it keeps the three-graph form (the 2-ranks-on-one-device test)
        one_graph = getattr(self, "dp_one_graph", True) and (
            self.dev.type != "cuda" or self.rccl is not None or
            (torch.distributed.is_initialized() and torch.distributed.get_backend(self.pg) == "nccl"))
        g_step = {}
        if one_graph:
            for a, b in pairs():
                (Bs, gb), (Bn, _) = sizes[a], sizes[b]
                g_step[(a, b)] = c.capture([[body_step(Bs, gb, Bn)]])[0]
        g_grad = g_adam = None
        if not one_graph:
            g_grad = [c.capture([[body_grad(Bs, gb)]])[0] for Bs, gb in sizes]
            g_adam = c.capture([[adam]])[0]
        self.dp_graphs_per_step = 1 if one_graph else 3

        def load_table(t):
            assert t.shape == self._g_table.shape
            self._g_table.copy_(t, non_blocking=True)
            nxt.copy_(t[0], non_blocking=True)
            g_front[1 if is_tail(0) else 0]()              # features of row 0
            cur.copy_(t[0], non_blocking=True)
            nxt.copy_(t[1 % rows], non_blocking=True)
            self._g_cursor.zero_()
            state["row"] = 0

        def replay():
            row = state["row"]
            self._bind()
            a, b = (1 if is_tail(row) else 0), (1 if is_tail((row + 1) % rows) else 0)
            if one_graph:
                g_step[(a, b)]()
            else:
                g_grad[a]()
                work = self.allreduce(async_op=True)
                g_front[b]()                               # MFCC of the next step while the collective is in flight
                if work is not None:
                    work.wait()                            # the caller's stream waits for the collective
                g_adam()                                   # ... and fetches the rows of the steps after
            state["row"] = (row + 1) % rows
            self.step_count += 1
            return self.loss
        load_table(table)
        return replay, load_table

    def _aux_stream(self):
        """The stream the next step's front-end is forked onto inside the captured data-parallel step."""
        if getattr(self, "_aux", None) is None:
            self._aux = torch.cuda.Stream(device=self.dev)
        return self._aux

    def set_lr(self, lr):
        self.lr = lr
        if getattr(self, "_g_lr", None) is not None:
            self._g_lr.fill_(float(lr))

    def step(self, image, pos, neg, global_batch=None):
        """One optimisation step on (image u8|f32 (B,3,H,H), pos, neg f32 (B,1,100,40)).  Asynchronous."""
        self._check(image, pos, neg)
        self.loss_and_grads(image, pos, neg, global_batch)
        self.allreduce()
        self.adam()
        return self.loss

    def step_from_dataset(self, images, image_index, pcm, clip_index, lens, global_batch=None):
        """One step with the data-loader work folded in (var_arm_loss_grad_pcm): sample b reads image row
        image_index[b] of the HBM-resident `images` (N,3,H,H) u8|f32; clips [pos | neg] read rows
        clip_index (2B) of `pcm` (M, n) int16 with lens (2B) valid samples (0 = "empty" class); the MFCC
        front-end runs inside the step.  Index tensors are int32 on the trainer's device."""
        flat = self.model.flat_parameters()
        B = image_index.numel()
        for t in (image_index, clip_index, lens):
            if t.dtype != torch.int32 or t.device != self.dev or not t.is_contiguous():
                raise VarHipError("index / length tensors must be contiguous int32 tensors on the trainer's device")
        if pcm.dtype != torch.int16 or clip_index.numel() != 2 * B or lens.numel() != 2 * B:
            raise VarHipError("pcm must be int16 and clip_index / lens must hold 2*B entries")
        if pcm.stride(0) % 2 or (pcm.dim() == 2 and pcm.stride(1) != 1):
            raise VarHipError("pcm rows must be contiguous with an even stride (the front-end loads sample pairs)")
        c = self.ctx
        c.ensure_plan(B, self.hw)
        self._bind()
        c.check(c.lib.var_arm_loss_grad_pcm(c.handle, c.stream(), ptr(flat), ptr(images),
                                            int(images.dtype == torch.uint8), images.stride(0), ptr(image_index),
                                            ptr(pcm), pcm.stride(0), ptr(clip_index), ptr(lens), B, self.hw,
                                            float(self.margin), 1.0 / self._gb(B, global_batch), ptr(self.gbuf),
                                            self._loss_ptr(), None), "var_arm_loss_grad_pcm")
        self.allreduce()
        self.adam()
        return self.loss

    def _inbatch_head(self, anchor, cand, target, tau, inv_count):
        c = self.ctx
        B, M = anchor.shape[0], cand.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        ga, gc = torch.empty_like(anchor), torch.empty_like(cand)
        scratch = torch.empty(2 * B, dtype=torch.float32, device=self.dev)
        c.check(c.lib.var_inbatch_loss_fwd_bwd(c.handle, c.stream(), ptr(anchor), ptr(cand), ptr(target), B, M, float(tau),
                                               float(inv_count), ptr(scratch), ptr(loss), ptr(ga), ptr(gc)),
                "var_inbatch_loss_fwd_bwd")
        return loss, ga, gc

    def step_inbatch(self, image, pos, neg, tau=0.1):
        """One optimisation step with the in-batch-negatives contrastive head instead of the triplet loss (BASELINE
        config 3; an extension, csrc/inbatch.hip): every rank's [positive ; negative] sound embeddings are all-gathered
        (6 KB per rank at B = 256), each rank scores its images against all of them, the candidate gradients are summed
        with one small all-reduce and each rank back-propagates its own rows; then the usual gradient all-reduce + Adam.
        Collectives go through torch.distributed, or through the C ABI when use_rccl() was called.  Equal shards."""
        self._check(image, pos, neg)
        c, m = self.ctx, self.model
        flat = m.flat_parameters()
        B = image.shape[0]
        c.ensure_plan(B, self.hw)
        dev = self.dev
        rccl, rank, world = self.rccl, self.rank, self.world
        stream = c.stream()
        self._bind()
        emb = torch.empty((3, B, 3), dtype=torch.float32, device=dev)          # [image | pos | neg]
        c.check(c.lib.var_arm_encoder_fwd(c.handle, stream, ptr(flat), ptr(image), int(image.dtype == torch.uint8),
                                          image.stride(0), ptr(pos), ptr(neg), B, self.hw, ptr(emb[0]), ptr(emb[1]), ptr(emb[2]),
                                          None, None, 1), "var_arm_encoder_fwd")
        local = emb[1:].reshape(2 * B, 3)
        if rccl is not None and world > 1:
            cand = rccl.allgather(local.reshape(-1)).view(-1, 3)
        elif world > 1:
            cand = torch.empty((world * 2 * B, 3), dtype=torch.float32, device=dev)
            torch.distributed.all_gather_into_tensor(cand, local.contiguous(), group=self.pg)
        else:
            cand = local
        target = torch.arange(B, dtype=torch.int32, device=dev) + rank * 2 * B
        loss, ga, gc = self._inbatch_head(emb[0], cand.contiguous(), target, tau, 1.0 / (B * world))
        if rccl is not None and world > 1:
            rccl.allreduce(gc.view(-1))
        elif world > 1:
            torch.distributed.all_reduce(gc, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        mine = gc[rank * 2 * B:(rank + 1) * 2 * B]
        c.check(c.lib.var_arm_encoder_bwd(c.handle, stream, ptr(flat), ptr(ga), ptr(mine[:B].contiguous()),
                                          ptr(mine[B:].contiguous()), ptr(self.gbuf)), "var_arm_encoder_bwd")
        self.gbuf[N_PARAMS:].copy_(loss)
        self.allreduce()
        self.adam()
        return self.loss

    def capture_inbatch_epoch_steps(self, images, pcm, batch, table, tau=0.1):
        """step_inbatch as a replayed step over the HBM-resident dataset (BASELINE configs[2]): per step the captured
        graphs gather the images of the current index row, run the MFCC front-end, the encoder forward, the
        in-batch-negatives head, the encoder backward and Adam + row fetch.  With one rank everything is ONE graph; with
        several the three collectives (all-gather of the candidates, all-reduce of the candidate gradients, all-reduce
        of the parameter gradients) stay eager between four graphs.  Returns (replay, load_table) like
        capture_epoch_steps; `table` rows are [image_index | clip_index (2B) | lens (2B)], equal shards."""
        B = batch
        rows, row_ints = int(table.shape[0]), int(table.shape[1])
        if row_ints != 5 * B or table.dtype != torch.int32 or table.device != self.dev or not table.is_contiguous():
            raise VarHipError("index table must be a contiguous int32 (rows, 5*batch) tensor on the trainer's device")
        c, dev, world, rank, rccl = self.ctx, self.dev, self.world, self.rank, self.rccl
        flat = self.model.flat_parameters()
        c.ensure_plan(B, self.hw)
        self._g_table = torch.empty_like(table)
        self._g_cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self._g_idx = torch.zeros(row_ints, dtype=torch.int32, device=dev)
        self._device_scalars(B)
        img = torch.zeros((B,) + tuple(images.shape[1:]), dtype=images.dtype, device=dev)
        feats = torch.zeros((2 * B, 1, 100, 40), dtype=torch.float32, device=dev)
        emb = torch.zeros((3, B, 3), dtype=torch.float32, device=dev)            # [image | pos | neg]
        cand = torch.zeros((world * 2 * B, 3), dtype=torch.float32, device=dev) if world > 1 else emb[1:].reshape(2 * B, 3)
        target = (torch.arange(B, dtype=torch.int32, device=dev) + rank * 2 * B).contiguous()
        loss1 = torch.zeros(1, dtype=torch.float32, device=dev)
        ga, gc = torch.zeros((B, 3), dtype=torch.float32, device=dev), torch.zeros_like(cand)
        scratch = torch.zeros(2 * B, dtype=torch.float32, device=dev)
        mine = gc[rank * 2 * B:(rank + 1) * 2 * B]
        clip_idx, lens = self._g_idx[B:3 * B], self._g_idx[3 * B:5 * B]
        self._keep_inbatch = (img, feats, emb, cand, target, loss1, ga, gc, scratch)

        def body_fwd():
            self._bind()
            torch.index_select(images, 0, self._g_idx[:B], out=img)     # (int32 indices: no widening launch in front)
            c.check(c.lib.var_mfcc(c.handle, c.stream(), ptr(pcm), ptr(lens), ptr(clip_idx), 2 * B, pcm.stride(0), 100,
                                   ptr(feats)), "var_mfcc")
            c.check(c.lib.var_arm_encoder_fwd(c.handle, c.stream(), ptr(flat), ptr(img), int(img.dtype == torch.uint8),
                                              img.stride(0), ptr(feats), ptr(feats[B:]), B, self.hw, ptr(emb[0]), ptr(emb[1]),
                                              ptr(emb[2]), None, None, 1), "var_arm_encoder_fwd")

        def body_loss():
            c.check(c.lib.var_inbatch_loss_fwd_bwd(c.handle, c.stream(), ptr(emb[0]), ptr(cand), ptr(target), B, cand.shape[0],
                                                   float(tau), 1.0 / (B * world), ptr(scratch), ptr(loss1), ptr(ga), ptr(gc)),
                    "var_inbatch_loss_fwd_bwd")

        def body_bwd():
            self._bind()
            c.check(c.lib.var_arm_encoder_bwd(c.handle, c.stream(), ptr(flat), ptr(ga), ptr(mine[:B]), ptr(mine[B:]),
                                              ptr(self.gbuf)), "var_arm_encoder_bwd")
            torch.mul(loss1, 1.0, out=self.gbuf[N_PARAMS:])          # (a kernel, not a memcpy node: _lib.new_graph)

        adam = self._body_adam(self._g_table, rows, row_ints, 0)
        body_fwd()                                                  # warm-up outside capture (torch's lazy initialisations)
        if world > 1:
            g_fwd, g_loss, g_bwd, g_adam = c.capture([[body_fwd], [body_loss], [body_bwd], [adam]])
        else:
            (g_all,) = c.capture([[body_fwd, body_loss, body_bwd, adam]])

        def load_table(t):
            assert t.shape == self._g_table.shape
            self._g_table.copy_(t, non_blocking=True)
            self._g_idx.copy_(t[0], non_blocking=True)
            self._g_cursor.zero_()

        def replay():
            self._bind()
            if world == 1:
                g_all()
            else:
                g_fwd()
                local = emb[1:].reshape(2 * B, 3)
                if rccl is not None:
                    cand.copy_(rccl.allgather(local.reshape(-1)).view(-1, 3))
                else:
                    torch.distributed.all_gather_into_tensor(cand, local, group=self.pg)
                g_loss()
                if rccl is not None:
                    rccl.allreduce(gc.view(-1))
                else:
                    torch.distributed.all_reduce(gc, op=torch.distributed.ReduceOp.SUM, group=self.pg)
                g_bwd()
                self.allreduce()
                g_adam()
            self.step_count += 1
            return self.loss
        load_table(table)
        return replay, load_table

    def step_from_pcm(self, image, pcm, lens, global_batch=None):
        """Same without gathering: image (B,3,H,H), pcm int16 (2B, n) = [pos | neg], lens (2B) (0 = empty)."""
        B = image.shape[0]
        idx = torch.arange(B, dtype=torch.int32, device=image.device)
        cidx = torch.arange(2 * B, dtype=torch.int32, device=image.device)
        return self.step_from_dataset(image, idx, pcm, cidx, lens, global_batch)

    def _check(self, image, pos, neg):
        for t in (image, pos, neg):
            if t is None or t.device != self.dev or not t.is_contiguous():
                raise VarHipError("VARTrainer.step needs contiguous tensors (image, pos, neg) on the trainer's device")
        if pos.dtype != torch.float32 or neg.dtype != torch.float32 or image.dtype not in (torch.uint8, torch.float32):
            raise VarHipError("image must be u8/f32 and MFCC f32")


def _save_outputs(tr, model, save_dir, fname, log):
    """Checkpoints and progress.csv are written by rank 0 only (every replica holds the same parameters; concurrent
    torch.save calls on one path can leave a corrupt file), with a barrier behind the write."""
    rank = getattr(tr, "rank", 0)
    dist = getattr(tr, "world", 1) > 1 and torch.distributed.is_available() and torch.distributed.is_initialized()
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
        torch.save(model.state_dict(), fname, _use_new_zipfile_serialization=False)   # legacy format, pretext_VAR.py:79
        log('Model saved to ' + fname)
    if dist:
        torch.distributed.barrier(group=getattr(tr, "pg", None))


def _save_progress(tr, save_dir, loss_list):
    if getattr(tr, "rank", 0) != 0:
        return
    os.makedirs(save_dir, exist_ok=True)
    with open(os.path.join(save_dir, 'progress.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['avg_loss'])
        for v in loss_list:
            w.writerow([v])


def train_representation(model, batches, epochs, lr=1e-4, weight_decay=1e-6, milestones=None, gamma=0.2,
                         margin=1.0, save_dir=None, save_interval=10, start_ep=0, log=print):
    """The loop of VAR/pretext_VAR.py:44-91.  `batches()` yields (image, sound_positive, sound_negative, gt)
    CUDA tensors for one epoch -- every batch, including a short last one (the reference's DataLoader has
    drop_last=False).  `milestones` defaults to the model's reference config: [10, 30, 50] for the Kuka model
    (fourInARow/config.py:43-44), [20, 30] for the iTHOR model (Envs/ai2thor/config.py:47-48).
    Returns the per-epoch average losses."""
    from .ithor import IthorTrainer, IthorVARPretextNet
    is_ithor = isinstance(model, IthorVARPretextNet)
    if milestones is None:
        milestones = (20, 30) if is_ithor else (10, 30, 50)
    trainer_cls = IthorTrainer if is_ithor else VARTrainer
    tr = trainer_cls(model, lr=lr, weight_decay=weight_decay, margin=margin)
    model.train()
    loss_list = []
    for ep in range(epochs):
        tr.lr = multistep_lr(lr, milestones, gamma, ep)
        losses = []
        for image, sp, sn, _gt in batches():
            tr.step(image.contiguous(), sp.float().contiguous(), sn.float().contiguous())
            losses.append(tr.loss.clone())
        stack = torch.stack(losses).reshape(-1)
        if is_ithor and model.gru_status():
            # bf16 mode: a persistent GRU launch did not get its whole grid resident (another process on the GPU) and timed
            # out.  Those steps carry a NaN loss and were SKIPPED by the optimiser (csrc/pack_adam.hip: Adam's guard word),
            # so parameters and moments are intact: take the per-step launches from here on and average the steps that ran.
            bad = int((~torch.isfinite(stack)).sum().item())
            log(f'persistent GRU launch timed out in {bad} step(s) of epoch {start_ep + ep} (status '
                f'{model.gru_status():#x}): skipped by the optimiser; switching to per-step GRU launches')
            model.set_gru_sequence(False)
            model._ensure_plan(tr.ctx, int(image.shape[0]))                # applies the form and clears the status words
            stack = stack[torch.isfinite(stack)]
            if stack.numel() == 0:
                raise VarHipError("every step of the epoch timed out in the persistent GRU launches")
        avg = float(stack.sum().item() / stack.numel())                   # np.sum(loss_ep)/len(loss_ep), :82
        loss_list.append(avg)
        log('average loss', avg)
        if save_dir and ((ep + 1) % save_interval == 0 or ep + 1 == epochs):
            _save_outputs(tr, model, save_dir, os.path.join(save_dir, str(start_ep + ep) + '.pt'), log)
    if save_dir:
        _save_progress(tr, save_dir, loss_list)
    model.eval()
    return loss_list


def train_representation_from_pool(model, pool, epochs, batch, lr=1e-4, weight_decay=1e-6, milestones=None,
                                   gamma=0.2, margin=1.0, save_dir=None, save_interval=10, start_ep=0, log=print,
                                   drop_last=False, _ctx=None):
    """The loop of VAR/pretext_VAR.py:44-91 over a TripletPool resident in HBM (u8 images, int16 clips, frozen pairs =
    VARFineTuneDataset, dataset.py:94-133): every epoch is a freshly shuffled index table (DataLoader(shuffle=True),
    drop_last=False incl. the short last batch) walked by the replayed step -- gather, MFCC front-end, forward, triplet
    loss, backward, Adam in one graph launch per step; MultiStepLR per epoch (utils.py:42-46); average loss per epoch =
    sum of the step losses / number of steps (:82), accumulated on the device (no per-step host sync); legacy-format
    checkpoints every `save_interval` epochs and at the end (:75-80), progress.csv (:87-91).
    Both models: the Kuka one (1 s clips, torchaudio-flavour MFCC, milestones [10, 30, 50]) and the iTHOR one (clips of up
    to 6 s, python_speech_features MFCC of 600 frames, milestones [20, 30] -- its reference default of 500 triplets at batch
    128 runs as 128 / 128 / 128 / 116, Envs/ai2thor/config.py:24,41-48)."""
    from .ithor import IthorTrainer, IthorVARPretextNet
    is_ithor = isinstance(model, IthorVARPretextNet)
    if milestones is None:
        milestones = (20, 30) if is_ithor else (10, 30, 50)
    if is_ithor:
        tr = IthorTrainer(model, lr=lr, weight_decay=weight_decay, margin=margin)
    else:
        tr = VARTrainer(model, lr=lr, weight_decay=weight_decay, margin=margin, _ctx=_ctx)
    if getattr(pool, "clip_tab", None) is None:
        pool.freeze_pairs()
    model.train()
    spe, bt = pool.steps_per_epoch(batch, drop_last), pool.tail_batch(batch, drop_last)
    if spe < 1:
        raise VarHipError(f"the pool holds {pool.n_items} triplets: no step of batch {batch} with drop_last={drop_last}")
    replay = load_table = None
    loss_list = []
    acc = torch.zeros(1, dtype=torch.float32, device=tr.dev)
    ran = torch.zeros(1, dtype=torch.float32, device=tr.dev)
    join_timeouts0 = 0 if is_ithor else tr.ctx.join_timeouts()
    for ep in range(epochs):
        tr.set_lr(multistep_lr(lr, milestones, gamma, ep))
        table = pool.epoch_index_table(batch, drop_last)
        if replay is None:
            replay, load_table = tr.capture_epoch_steps(pool.images, pool.clips, batch, table, steps_per_epoch=spe,
                                                        tail_batch=bt)
        else:
            load_table(table)
        acc.zero_()
        ran.zero_()
        for _ in range(spe):
            l = replay()
            ok = torch.isfinite(l)                         # (device-side: a timed-out iTHOR step reads NaN and was skipped by Adam)
            acc += torch.where(ok, l, torch.zeros_like(l)).reshape(-1)[:1]
            ran += ok.reshape(-1)[:1].float()
        n_ran = int(ran.item())
        if not is_ithor and tr.ctx.join_timeouts() > join_timeouts0:
            # a replayed step hands over between its two streams on the device (csrc/heads.hip); a wait that gave up means a
            # branch of some step never ran -- a fault of the device or the runtime, and that step's numbers were garbage
            raise VarHipError(f"a device-side stream hand-over timed out in epoch {start_ep + ep} (var_join_status: "
                              f"{tr.ctx.join_timeouts() - join_timeouts0}); var_set_streams(ctx, 3 | 64) restores the stream edges")
        if is_ithor and n_ran < spe:
            # bf16 mode: a persistent GRU launch did not get its whole grid resident and timed out; the optimiser skipped those
            # steps (the Adam kernels' guards, csrc/pack_adam.hip), parameters and moments are intact.  The captured graphs keep
            # the persistent form: switch to per-step GRU launches and capture again for the next epoch.
            log(f'persistent GRU launch timed out in {spe - n_ran} step(s) of epoch {start_ep + ep} (status '
                f'{model.gru_status():#x}): skipped by the optimiser; switching to per-step GRU launches')
            model.set_gru_sequence(False)
            model._ensure_plan(tr.ctx, int(batch))         # applies the form and clears the status words
            replay = None
            if n_ran == 0:
                raise VarHipError("every step of the epoch timed out in the persistent GRU launches")
        avg = float(acc.item()) / max(n_ran, 1)               # np.sum(loss_ep)/len(loss_ep), :82, over the steps that ran
        loss_list.append(avg)
        log('average loss', avg)
        if save_dir and ((ep + 1) % save_interval == 0 or ep + 1 == epochs):
            _save_outputs(tr, model, save_dir, os.path.join(save_dir, str(start_ep + ep) + '.pt'), log)
    if save_dir:
        _save_progress(tr, save_dir, loss_list)
    model.eval()
    return loss_list
