"""Counterpart of VAR_Pretext.trainRepresentation (VAR/pretext_VAR.py:16-95): the step body
zero_grad -> forward -> TripletMarginLoss -> backward -> Adam.step as fused HIP launches on flat
arenas, per-epoch MultiStepLR (utils.py:42-46), periodic legacy-format checkpoints and
progress.csv.  Data parallel: one process per GPU, one RCCL all-reduce of the flat gradient
arena (+ the loss scalar riding in its last slot) per step."""
import csv
import os

import torch

from ._lib import Context, VarHipError, current_stream_handle, ptr
from .layout import N_PARAMS
from .ops import mfcc as mfcc_op


def multistep_lr(base_lr, milestones, gamma, epoch):
    """lr in effect during 0-based `epoch` under MultiStepLR(milestones, gamma) stepped once per epoch."""
    lr = base_lr
    for ms in sorted(milestones):
        if epoch >= ms:
            lr *= gamma
    return lr


class VARTrainer:
    def __init__(self, model, lr=1e-4, weight_decay=1e-6, betas=(0.9, 0.999), eps=1e-8, margin=1.0,
                 process_group=None):
        flat = model.flat_parameters()
        if not flat.is_cuda:
            raise VarHipError("VARTrainer needs the model on a GPU (no CPU fallback)")
        self.model = model
        self.dev = flat.device
        self.ctx = Context.get(self.dev.index)
        self.lr, self.wd, self.betas, self.eps, self.margin = lr, weight_decay, betas, eps, margin
        self.hw = model.config.img_dim[1]
        # gradient arena with one extra slot for the loss so that ONE all-reduce carries both
        self.gbuf = torch.zeros(N_PARAMS + 1, dtype=torch.float32, device=self.dev)
        self.exp_avg = torch.zeros(N_PARAMS, dtype=torch.float32, device=self.dev)
        self.exp_avg_sq = torch.zeros(N_PARAMS, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self.pg = process_group
        self.world = 1
        # VAR_FORCE_ALLREDUCE=1: run the data-parallel code path (RCCL all-reduce between two graphs) even with
        # one rank, to rehearse the multi-GPU step on a single-GPU box
        self.force_collective = os.environ.get("VAR_FORCE_ALLREDUCE") == "1"
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.pack()

    def pack(self):
        """Refresh the packed weight images from the parameter arena (needed after loading a checkpoint or any
        direct edit of the parameters; every optimiser step does it by itself)."""
        c = self.ctx
        c.check(c.lib.var_pack_weights(c.handle, current_stream_handle(), ptr(self.model.flat_parameters())),
                "var_pack_weights")

    @property
    def grads(self):
        return self.gbuf[:N_PARAMS]

    @property
    def loss(self):
        """Device scalar: the (global) mean triplet loss of the last step; reading it syncs."""
        return self.gbuf[N_PARAMS:]

    def loss_and_grads(self, image, pos, neg, global_batch=None):
        """fwd + loss + bwd into the gradient arena (no optimiser step, no collective)."""
        flat = self.model.flat_parameters()
        B = image.shape[0]
        gb = B * self.world if global_batch is None else global_batch
        c = self.ctx
        c.ensure_plan(B, self.hw)
        c.check(c.lib.var_arm_loss_grad(c.handle, current_stream_handle(), ptr(flat), ptr(image),
                                        int(image.dtype == torch.uint8), image.stride(0), ptr(pos), ptr(neg),
                                        B, self.hw, float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                        self.gbuf.data_ptr() + 4 * N_PARAMS, None), "var_arm_loss_grad")

    def use_rccl(self, comm):
        """Route the gradient all-reduce through the C ABI (comm.RcclComm, var_allreduce_grads) instead of
        torch.distributed; `comm.size` becomes the world size of the loss / gradient scaling."""
        self.rccl = comm
        self.world = comm.size
        return self

    def allreduce(self):
        if getattr(self, "rccl", None) is not None:
            self.rccl.allreduce(self.gbuf)
        elif self.world > 1 or (self.force_collective and torch.distributed.is_initialized()):
            torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def adam(self):
        self.step_count += 1
        flat = self.model.flat_parameters()
        c = self.ctx
        c.check(c.lib.var_adam_step(c.handle, current_stream_handle(), ptr(flat), ptr(self.gbuf),
                                    ptr(self.exp_avg), ptr(self.exp_avg_sq), N_PARAMS, float(self.lr),
                                    float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                                    int(self.step_count)), "var_adam_step")

    # ---- HIP-graph replay of the whole step (launch-bound otherwise: ~35 kernels + stream fork/joins) ----
    def capture_dataset_step(self, images, pcm, batch, global_batch=None, _table=None):
        """Capture step_from_dataset(images, idx, pcm, clip_idx, lens) once; returns replay(idx_row) where
        idx_row is an int32 CUDA tensor of 5*batch entries [image_index | clip_index (2B) | lens (2B)].
        Step count and learning rate live on the device (var_adam_step_dev); set_lr() updates the latter."""
        dev = self.dev
        B = batch
        self._g_idx = torch.zeros(5 * B, dtype=torch.int32, device=dev)
        self._g_lr = torch.full((1,), float(self.lr), dtype=torch.float32, device=dev)
        self._g_step = torch.full((1,), int(self.step_count), dtype=torch.int32, device=dev)
        flat = self.model.flat_parameters()
        c = self.ctx
        c.ensure_plan(B, self.hw)
        gb = B * self.world if global_batch is None else global_batch
        img_idx, clip_idx, lens = self._g_idx[:B], self._g_idx[B:3 * B], self._g_idx[3 * B:]

        def body_grad():
            c.check(c.lib.var_arm_loss_grad_pcm(c.handle, current_stream_handle(), ptr(flat), ptr(images),
                                                int(images.dtype == torch.uint8), images.stride(0), ptr(img_idx),
                                                ptr(pcm), pcm.stride(0), ptr(clip_idx), ptr(lens), B, self.hw,
                                                float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                                self.gbuf.data_ptr() + 4 * N_PARAMS, None), "var_arm_loss_grad_pcm")

        def body_adam():
            tab, rows, row_ints = _table if _table is not None else (None, 0, 0)
            c.check(c.lib.var_adam_step_graph(c.handle, current_stream_handle(), ptr(flat), ptr(self.gbuf),
                                              ptr(self.exp_avg), ptr(self.exp_avg_sq), N_PARAMS, ptr(self._g_lr),
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              float(self.wd), ptr(self._g_step),
                                              ptr(tab) if tab is not None else None, row_ints, rows,
                                              ptr(self._g_cursor) if tab is not None else None,
                                              ptr(self._g_idx) if tab is not None else None, 0), "var_adam_step_graph")

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        graphs = []
        with torch.cuda.stream(side):
            if self.world > 1 or self.force_collective:   # the RCCL all-reduce stays eager between two graphs
                for body in (body_grad, body_adam):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        body()
                    graphs.append(g)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    body_grad()
                    body_adam()
                graphs.append(g)
        torch.cuda.current_stream().wait_stream(side)

        def replay(idx_row):
            if idx_row is not None:                      # None: the captured step walks its own index table
                self._g_idx.copy_(idx_row, non_blocking=True)
            graphs[0].replay()
            if len(graphs) > 1:
                self.allreduce()
                graphs[1].replay()
            self.step_count += 1
            return self.loss
        return replay

    def capture_epoch_steps(self, images, pcm, batch, table, global_batch=None):
        """Like capture_dataset_step, with the data-loader cursor on the device too: `table` is an int32 CUDA
        tensor (rows, 5*batch) of step rows [image_index | clip_index (2B) | lens (2B)] (one or more shuffled epochs,
        SyntheticTripletPool.index_table).  Returns (replay, load_table): replay() launches the captured step --
        no host-side copy, the step itself fetches the next row (var_adam_step_graph); load_table(t) installs a
        new table of the same shape (next epochs) and rewinds the cursor.  After `rows` replays without a new
        table the walk starts over."""
        dev = self.dev
        B = batch
        rows, row_ints = int(table.shape[0]), int(table.shape[1])
        assert row_ints == 5 * B and table.dtype == torch.int32 and table.is_cuda and table.is_contiguous()
        self._g_table = torch.empty_like(table)
        self._g_cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.world > 1 or self.force_collective:
            return self._capture_epoch_steps_dp(images, pcm, B, table, global_batch)
        replay_row = self.capture_dataset_step(images, pcm, B, global_batch, _table=(self._g_table, rows, row_ints))

        def load_table(t):
            assert t.shape == self._g_table.shape
            self._g_table.copy_(t, non_blocking=True)
            self._g_idx.copy_(t[0], non_blocking=True)
            self._g_cursor.zero_()

        load_table(table)

        def replay():
            return replay_row(None)
        return replay, load_table

    def _capture_epoch_steps_dp(self, images, pcm, B, table, global_batch):
        """Data-parallel form of capture_epoch_steps.  The gradient all-reduce (RCCL, eager between graphs) has
        nothing to overlap with inside the step -- every gradient is complete only at the end of the backward -- but
        the audio front-end of the NEXT step does not depend on the weights: per step
            graph [gather + fwd + loss + bwd, MFCC features of this step precomputed]
            -> all_reduce(async) || MFCC of the next step (graph, on the caller's stream)
            -> wait -> graph [Adam + re-pack]
        so up to an MFCC kernel's worth (60 us) of collective latency is hidden."""
        dev = self.dev
        c = self.ctx
        flat = self.model.flat_parameters()
        c.ensure_plan(B, self.hw)
        rows = int(table.shape[0])
        gb = B * self.world if global_batch is None else global_batch
        self._g_idx = torch.zeros(5 * B, dtype=torch.int32, device=dev)
        self._g_lr = torch.full((1,), float(self.lr), dtype=torch.float32, device=dev)
        self._g_step = torch.full((1,), int(self.step_count), dtype=torch.int32, device=dev)
        self._g_mfcc = torch.zeros(2 * B, 1, 100, 40, dtype=torch.float32, device=dev)
        img_idx, clip_idx, lens = self._g_idx[:B], self._g_idx[B:3 * B], self._g_idx[3 * B:]

        def body_grad():
            c.check(c.lib.var_arm_loss_grad_gather(c.handle, current_stream_handle(), ptr(flat), ptr(images),
                                                   int(images.dtype == torch.uint8), images.stride(0), ptr(img_idx),
                                                   ptr(self._g_mfcc), ptr(self._g_mfcc[B:]), B, self.hw,
                                                   float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                                   self.gbuf.data_ptr() + 4 * N_PARAMS, None), "var_arm_loss_grad_gather")

        def body_front():
            c.check(c.lib.var_mfcc(c.handle, current_stream_handle(), ptr(pcm), ptr(lens), ptr(clip_idx), 2 * B,
                                   pcm.stride(0), 100, ptr(self._g_mfcc)), "var_mfcc")

        def body_adam():
            c.check(c.lib.var_adam_step_graph(c.handle, current_stream_handle(), ptr(flat), ptr(self.gbuf),
                                              ptr(self.exp_avg), ptr(self.exp_avg_sq), N_PARAMS, ptr(self._g_lr),
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              float(self.wd), ptr(self._g_step), ptr(self._g_table), 5 * B, rows,
                                              ptr(self._g_cursor), ptr(self._g_idx), B), "var_adam_step_graph")

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        graphs = []
        with torch.cuda.stream(side):
            body_front()                                   # warm-up outside capture
            for body in (body_grad, body_front, body_adam):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    body()
                graphs.append(g)
        torch.cuda.current_stream().wait_stream(side)
        g_grad, g_front, g_adam = graphs
        collective = self.world > 1 or (self.force_collective and torch.distributed.is_initialized())

        def load_table(t):
            assert t.shape == self._g_table.shape
            self._g_table.copy_(t, non_blocking=True)
            self._g_idx.copy_(t[0], non_blocking=True)
            g_front.replay()                               # features of row 0
            self._g_idx[B:].copy_(t[1 % rows][B:], non_blocking=True)   # clip entries run one step ahead
            self._g_cursor.zero_()

        load_table(table)

        def replay():
            g_grad.replay()
            work = None
            if collective:
                work = torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                    async_op=True)
            g_front.replay()                               # MFCC of the next step while the collective is in flight
            if work is not None:
                work.wait()                                # the caller's stream waits for the collective
            g_adam.replay()                                # ... and fetches the rows of the steps after
            self.step_count += 1
            return self.loss
        return replay, load_table

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
        front-end runs inside the step.  Index tensors are int32 CUDA."""
        flat = self.model.flat_parameters()
        B = image_index.numel()
        gb = B * self.world if global_batch is None else global_batch
        for t in (image_index, clip_index, lens):
            if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
                raise VarHipError("index / length tensors must be contiguous int32 CUDA tensors")
        if pcm.dtype != torch.int16 or clip_index.numel() != 2 * B or lens.numel() != 2 * B:
            raise VarHipError("pcm must be int16 and clip_index / lens must hold 2*B entries")
        c = self.ctx
        c.ensure_plan(B, self.hw)
        c.check(c.lib.var_arm_loss_grad_pcm(c.handle, current_stream_handle(), ptr(flat), ptr(images),
                                            int(images.dtype == torch.uint8), images.stride(0), ptr(image_index),
                                            ptr(pcm), pcm.stride(0), ptr(clip_index), ptr(lens), B, self.hw,
                                            float(self.margin), 1.0 / gb, ptr(self.gbuf),
                                            self.gbuf.data_ptr() + 4 * N_PARAMS, None), "var_arm_loss_grad_pcm")
        self.allreduce()
        self.adam()
        return self.loss

    def step_inbatch(self, image, pos, neg, tau=0.1):
        """One optimisation step with the in-batch-negatives contrastive head instead of the triplet loss (BASELINE
        config 3; an extension, csrc/inbatch.hip): every rank's [positive ; negative] sound embeddings are all-gathered
        (9 KB per rank at B = 256), each rank scores its images against all of them, the candidate gradients are summed
        with one small all-reduce and each rank back-propagates its own rows; then the usual gradient all-reduce + Adam.
        Collectives go through torch.distributed, or through the C ABI when use_rccl() was called."""
        from .ops import inbatch_contrastive_loss
        self._check(image, pos, neg)
        c, m = self.ctx, self.model
        flat = m.flat_parameters()
        B = image.shape[0]
        c.ensure_plan(B, self.hw)
        dev = self.dev
        rccl = getattr(self, "rccl", None)
        rank = rccl.rank if rccl is not None else (torch.distributed.get_rank(self.pg) if self.world > 1 else 0)
        stream = current_stream_handle()
        emb = torch.empty((3, B, 3), dtype=torch.float32, device=dev)          # [image | pos | neg]
        c.check(c.lib.var_arm_encoder_fwd(c.handle, stream, ptr(flat), ptr(image), int(image.dtype == torch.uint8),
                                          image.stride(0), ptr(pos), ptr(neg), B, self.hw, ptr(emb[0]), ptr(emb[1]), ptr(emb[2]),
                                          None, None, 1), "var_arm_encoder_fwd")
        local = emb[1:].reshape(2 * B, 3)
        if rccl is not None and self.world > 1:
            cand = rccl.allgather(local.reshape(-1)).view(-1, 3)
        elif self.world > 1:
            cand = torch.empty((self.world * 2 * B, 3), dtype=torch.float32, device=dev)
            torch.distributed.all_gather_into_tensor(cand, local.contiguous(), group=self.pg)
        else:
            cand = local
        target = torch.arange(B, dtype=torch.int32, device=dev) + rank * 2 * B
        loss, ga, gc = inbatch_contrastive_loss(emb[0], cand, target, tau=tau, inv_count=1.0 / (B * self.world))
        if rccl is not None and self.world > 1:
            rccl.allreduce(gc.view(-1))
        elif self.world > 1:
            torch.distributed.all_reduce(gc, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        mine = gc[rank * 2 * B:(rank + 1) * 2 * B]
        c.check(c.lib.var_arm_encoder_bwd(c.handle, stream, ptr(flat), ptr(ga), ptr(mine[:B].contiguous()),
                                          ptr(mine[B:].contiguous()), ptr(self.gbuf)), "var_arm_encoder_bwd")
        self.gbuf[N_PARAMS:].copy_(loss)
        self.allreduce()
        self.adam()
        return self.loss

    def step_from_pcm(self, image, pcm, lens, global_batch=None):
        """Same without gathering: image (B,3,H,H), pcm int16 (2B, n) = [pos | neg], lens (2B) (0 = empty)."""
        B = image.shape[0]
        idx = torch.arange(B, dtype=torch.int32, device=image.device)
        cidx = torch.arange(2 * B, dtype=torch.int32, device=image.device)
        return self.step_from_dataset(image, idx, pcm, cidx, lens, global_batch)

    def _check(self, image, pos, neg):
        for t in (image, pos, neg):
            if t is None or not t.is_cuda or not t.is_contiguous():
                raise VarHipError("VARTrainer.step needs contiguous CUDA tensors (image, pos, neg)")
        if pos.dtype != torch.float32 or neg.dtype != torch.float32 or image.dtype not in (torch.uint8, torch.float32):
            raise VarHipError("image must be u8/f32 and MFCC f32")


def train_representation(model, batches, epochs, lr=1e-4, weight_decay=1e-6, milestones=(10, 30, 50), gamma=0.2,
                         margin=1.0, save_dir=None, save_interval=10, start_ep=0, log=print):
    """The loop of VAR/pretext_VAR.py:44-91.  `batches()` yields (image, sound_positive, sound_negative, gt)
    CUDA tensors for one epoch.  Returns the per-epoch average losses."""
    from .ithor import IthorTrainer, IthorVARPretextNet
    trainer_cls = IthorTrainer if isinstance(model, IthorVARPretextNet) else VARTrainer
    tr = trainer_cls(model, lr=lr, weight_decay=weight_decay, margin=margin)
    model.train()
    loss_list = []
    for ep in range(epochs):
        tr.lr = multistep_lr(lr, milestones, gamma, ep)
        losses = []
        for image, sp, sn, _gt in batches():
            tr.step(image.contiguous(), sp.float().contiguous(), sn.float().contiguous())
            losses.append(tr.loss.clone())
        avg = float(torch.stack(losses).sum().item() / len(losses))      # np.sum(loss_ep)/len(loss_ep), :82
        loss_list.append(avg)
        log('average loss', avg)
        if save_dir and ((ep + 1) % save_interval == 0 or ep + 1 == epochs):
            os.makedirs(save_dir, exist_ok=True)
            fname = os.path.join(save_dir, str(start_ep + ep) + '.pt')
            torch.save(model.state_dict(), fname, _use_new_zipfile_serialization=False)   # legacy format, :79
            log('Model saved to ' + fname)
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, 'progress.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['avg_loss'])
            for v in loss_list:
                w.writerow([v])
    model.eval()
    return loss_list
