"""ctypes binding of libvar_hip.so (include/var_hip.h).  Fails loudly when the library is
missing or a call returns an error -- there is no fallback path."""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAR_HIP_LIB: another build of the same library (A/B timing of two builds in one GPU session)
_LIB_PATH = os.environ.get("VAR_HIP_LIB") or os.path.join(_HERE, "libvar_hip.so")
_lib = None
_lock = threading.Lock()


class VarHipError(RuntimeError):
    pass


def library_path():
    return _LIB_PATH


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 into libvar_hip.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return _LIB_PATH


_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

_SIGNATURES = {
    "var_init": (_i, [_i, ctypes.POINTER(_vp)]),
    "var_destroy": (_i, [_vp]),
    "var_last_error": (ctypes.c_char_p, [_vp]),
    "var_param_count": (_i, []),
    "var_plan": (_i, [_vp, _i, _i]),
    "var_plan_generation": (_i, [_vp]),
    "var_saved_generation": (_i, [_vp]),
    "var_weights_create": (_i, [_vp, ctypes.POINTER(_vp)]),
    "var_weights_destroy": (_i, [_vp, _vp]),
    "var_weights_bind": (_i, [_vp, _vp]),
    "var_pack_weights": (_i, [_vp, _vp, _vp]),
    "var_arm_encoder_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i]),
    "var_arm_encoder_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "var_row_dot": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "var_triplet_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "var_arm_loss_grad": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "var_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i]),
    "var_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _f, _f, _f, _f, _vp]),
    "var_adam_step_graph": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _f, _f, _f, _f, _vp, _vp, _i, _i, _vp, _vp, _i]),
    "var_mfcc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "var_mfcc_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "var_arm_loss_grad_pcm": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "var_arm_loss_grad_gather": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "var_inbatch_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "var_comm_unique_id": (_i, [_vp, _vp]),
    "var_comm_init": (_i, [_vp, _i, _i, _vp]),
    "var_comm_destroy": (_i, [_vp]),
    "var_allreduce_grads": (_i, [_vp, _vp, _vp, _l]),
    "var_allgather_emb": (_i, [_vp, _vp, _vp, _vp, _l]),
    "var_armnet_param_count": (_i, []),
    "var_armnet_plan": (_i, [_vp, _i]),
    "var_armnet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "var_armnet_status": (_i, [_vp, _vp]),
    "var_join_status": (_i, [_vp, _vp]),
    "var_set_reward_dot": (_i, [_vp, _vp, _vp]),
    "var_armnet_clear_status": (_i, [_vp]),
    "var_debug_armnet_drop_workgroup": (_i, [_vp]),
    "var_mfcc_psf": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "var_ithor_param_count": (_i, []),
    "var_ithor_plan": (_i, [_vp, _i, _i]),
    "var_ithor_set_bf16": (_i, [_vp, _i]),
    "var_ithor_set_gru_sequence": (_i, [_vp, _i]),
    "var_ithor_gru_status": (_i, [_vp, _vp]),
    "var_ithor_guard_loss": (_i, [_vp, _vp]),
    "var_debug_ithor_gru_drop_workgroup": (_i, [_vp]),
    "var_ithor_encoder_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i]),
    "var_ithor_saved_generation": (_i, [_vp]),
    "var_ithor_encoder_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "var_ithor_loss_grad": (_i, [_vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "var_profile_tag_count": (_i, []),
    "var_profile_tag_name": (ctypes.c_char_p, [_i]),
    "var_profile_select": (_i, [_vp, _i]),
    "var_set_streams": (_i, [_vp, _i]),
    "var_profile_read": (_i, [_vp, ctypes.POINTER(_f), ctypes.POINTER(_i)]),
    "var_debug_buffer": (_i, [_vp, ctypes.c_char_p, ctypes.POINTER(_vp), ctypes.POINTER(_l)]),
    "var_debug_ithor_dense": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


def load_library():
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise VarHipError(
                    f"{_LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                    "(there is no CPU/eager fallback for the VAR hot path)")
            lib = ctypes.CDLL(_LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def new_graph():
    """The torch.cuda.CUDAGraph of the package's captures (one place to change how they are made).

    History, so that it is not repeated: captured iTHOR steps used to return loss == margin with all-zero gradients -- always
    from some replay after a device or stream synchronise, in about half the processes.  The initial GRU state, which the step
    zeroed with hipMemsetAsync, was then full of 16-byte patterns of host pointers: on this stack (ROCm 7.2 runtime under torch
    2.10, MI355X) a memset NODE of a captured graph can lose its fill pattern across a synchronise.  keep_graph=True and never
    destroying graphs hid it for a while (they moved the instantiation); the cure is that the library enqueues kernels only
    (csrc/var_common.h: var_zero_async / var_copy_async) -- plain graphs, destroyed whenever Python likes, are fine then
    (tools/graph_replay_check.py runs the sequences that used to fail)."""
    import torch
    return torch.cuda.CUDAGraph()


class Context:
    """One var_ctx per (process, device)."""
    _by_device = {}

    def __init__(self, device_index):
        self.lib = load_library()
        self.device_index = int(device_index)
        h = _vp()
        rc = self.lib.var_init(self.device_index, ctypes.byref(h))
        if rc != 0:
            raise VarHipError(f"var_init({device_index}) failed ({rc}): {self.lib.var_last_error(None).decode()}")
        self.handle = h
        self.plan = (0, 0)

    @classmethod
    def get(cls, device_index):
        ctx = cls._by_device.get(device_index)
        if ctx is None:
            ctx = cls._by_device[device_index] = Context(device_index)
        return ctx

    def check(self, rc, what):
        if rc != 0:
            raise VarHipError(f"{what} failed ({rc}): {self.lib.var_last_error(self.handle).decode()}")

    def stream(self):
        """The caller's current HIP stream as the void* the C ABI takes."""
        return current_stream_handle()

    def new_weights(self):
        return Weights(self)

    def capture(self, groups):
        """Capture each group of bodies (callables that enqueue C-ABI launches on the current stream) into one HIP
        graph; returns one replay callable per group."""
        import torch
        dev = torch.device("cuda", self.device_index)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graphs = []
        with torch.cuda.stream(side):
            for bodies in groups:
                g = new_graph()
                # thread-local capture: the RCCL watchdog thread of torch.distributed may query events meanwhile
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for body in bodies:
                        body()
                graphs.append(g)
        torch.cuda.current_stream(dev).wait_stream(side)
        return [g.replay for g in graphs]

    def ensure_plan(self, batch, hw):
        if self.plan[1] != hw or self.plan[0] < batch:
            self.check(self.lib.var_plan(self.handle, int(batch), int(hw)), "var_plan")
            self.plan = (int(batch), int(hw))

    def profile_select(self, tag):
        self.check(self.lib.var_profile_select(self.handle, int(tag)), "var_profile_select")

    def set_streams(self, mask):
        """Stream plan of a step (include/var_hip.h); -1 = default.  Returns the previous mask."""
        return self.lib.var_set_streams(self.handle, int(mask))

    def join_timeouts(self):
        """Device-side stream hand-overs of training steps that gave up since var_init (include/var_hip.h: var_join_status)."""
        n = ctypes.c_uint(0)
        self.check(self.lib.var_join_status(self.handle, ctypes.byref(n)), "var_join_status")
        return n.value

    def profile_read(self):
        ms, n = _f(), _i()
        self.check(self.lib.var_profile_read(self.handle, ctypes.byref(ms), ctypes.byref(n)), "var_profile_read")
        return ms.value, n.value

    def tag_names(self):
        return [self.lib.var_profile_tag_name(t).decode() for t in range(self.lib.var_profile_tag_count())]

    def debug_buffer(self, name):
        import torch
        p, n = _vp(), _l()
        self.check(self.lib.var_debug_buffer(self.handle, name.encode(), ctypes.byref(p), ctypes.byref(n)),
                   "var_debug_buffer")
        out = torch.empty(n.value, dtype=torch.float32, device=f"cuda:{self.device_index}")
        torch.cuda.synchronize()
        # raw D2D copy through a ctypes-wrapped view is not available; use hipMemcpy via torch's runtime
        rt = ctypes.CDLL("libamdhip64.so")
        rt.hipMemcpy.argtypes = [_vp, _vp, ctypes.c_size_t, _i]
        rc = rt.hipMemcpy(out.data_ptr(), p, n.value * 4, 3)   # hipMemcpyDeviceToDevice
        if rc != 0:
            raise VarHipError(f"hipMemcpy failed ({rc})")
        return out


class Weights:
    """Per-model packed weight image (var_weights_create / _bind / var_pack_weights): the kernels read conv filters
    re-laid for the matrix cores next to the parameter arena; each model owns its copy, so a training model and a
    frozen encoder can alternate on one device context."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = _vp()
        ctx.check(ctx.lib.var_weights_create(ctx.handle, ctypes.byref(h)), "var_weights_create")
        self.handle = h
        self.key = None                    # what the image was last packed from (model-side change detector)

    def bind(self):
        c = self.ctx
        c.check(c.lib.var_weights_bind(c.handle, self.handle), "var_weights_bind")

    def pack(self, flat, key=None):
        c = self.ctx
        self.bind()
        c.check(c.lib.var_pack_weights(c.handle, c.stream(), flat.data_ptr()), "var_pack_weights")
        self.key = key

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                self.ctx.lib.var_weights_destroy(self.ctx.handle, self.handle)
        except Exception:                  # interpreter shutdown
            pass
        self.handle = None


def ptr(t):
    return None if t is None else t.data_ptr()


def current_stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream
