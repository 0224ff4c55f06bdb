"""RCCL collectives through the C ABI (var_comm_* / var_allreduce_grads / var_allgather_emb, csrc/comm.hip) for hosts
that do not use torch.distributed.  The unique id travels over whatever channel the host has (a file, MPI, a socket,
or torch.distributed's store): rank 0 calls unique_id(), every rank calls init(rank, nranks, id)."""
import ctypes

import torch

from ._lib import Context, VarHipError, current_stream_handle, ptr


class RcclComm:
    def __init__(self, device_index=0):
        self.ctx = Context.get(device_index)
        self.rank, self.size = 0, 0

    def unique_id(self):
        buf = ctypes.create_string_buffer(128)
        self.ctx.check(self.ctx.lib.var_comm_unique_id(self.ctx.handle, ctypes.cast(buf, ctypes.c_void_p)), "var_comm_unique_id")
        return buf.raw

    def init(self, rank, nranks, unique_id):
        if len(unique_id) != 128:
            raise VarHipError("an RCCL unique id is 128 bytes")
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self.ctx.check(self.ctx.lib.var_comm_init(self.ctx.handle, int(rank), int(nranks), ctypes.cast(buf, ctypes.c_void_p)),
                       "var_comm_init")
        self.rank, self.size = int(rank), int(nranks)
        return self

    def allreduce(self, flat):
        """In-place sum over the ranks of a contiguous f32 CUDA tensor (the gradient arena + loss slot)."""
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise VarHipError("allreduce needs a contiguous f32 CUDA tensor")
        self.ctx.check(self.ctx.lib.var_allreduce_grads(self.ctx.handle, current_stream_handle(), ptr(flat), flat.numel()),
                       "var_allreduce_grads")
        return flat

    def allgather(self, local):
        """(size * n,) tensor whose slice r is rank r's `local` (n f32 values)."""
        if not (local.is_cuda and local.dtype == torch.float32 and local.is_contiguous()):
            raise VarHipError("allgather needs a contiguous f32 CUDA tensor")
        out = torch.empty(self.size * local.numel(), dtype=torch.float32, device=local.device)
        self.ctx.check(self.ctx.lib.var_allgather_emb(self.ctx.handle, current_stream_handle(), ptr(local), ptr(out),
                                                      local.numel()), "var_allgather_emb")
        return out

    def destroy(self):
        self.ctx.check(self.ctx.lib.var_comm_destroy(self.ctx.handle), "var_comm_destroy")
        self.size = 0
