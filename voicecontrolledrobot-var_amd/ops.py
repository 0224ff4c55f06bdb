"""Thin Python fronts of the stand-alone C-ABI operators (no arithmetic here)."""
import torch

from ._lib import Context, VarHipError, current_stream_handle, ptr


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VarHipError("the VAR hot path runs on the GPU only: got a CPU tensor (no CPU fallback)")


def triplet_margin_loss(anchor, positive, negative, margin=1.0, inv_count=None, want_grads=True):
    """torch.nn.TripletMarginLoss(margin, p=2) forward(+backward) in one HIP launch
    (VAR/pretext_VAR.py:38,64).  Returns (loss[1], ga, gp, gn)."""
    _require_cuda(anchor, positive, negative)
    a, p, n = (t.contiguous().float() for t in (anchor, positive, negative))
    B = a.shape[0]
    ctx = Context.get(a.device.index)
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    ga = torch.empty_like(a) if want_grads else None
    gp = torch.empty_like(a) if want_grads else None
    gn = torch.empty_like(a) if want_grads else None
    ctx.check(ctx.lib.var_triplet_fwd_bwd(ctx.handle, current_stream_handle(), ptr(a), ptr(p), ptr(n), B,
                                          float(margin), float(1.0 / B if inv_count is None else inv_count),
                                          ptr(loss), ptr(ga), ptr(gp), ptr(gn)), "var_triplet_fwd_bwd")
    return loss, ga, gp, gn


# STFT parameters per dataset at 16 kHz (Envs/audioLoader.py:23-31): (n_fft, win_length, hop_length)
DATASET_STFT = {"GoogleCommand": (512, 400, 160), "ESC50": (512, 400, 160), "FSC": (512, 400, 160),
                "Spatial": (512, 400, 160), "Synthetic": (512, 400, 160),
                "NSynth": (1024, 800, 640), "UrbanSound": (1024, 800, 640)}


def mfcc(pcm, lens=None, out_frames=100, clip_index=None, n_fft=512, win_length=400, hop_length=160, dataset=None):
    """int16 PCM (nclips, nsamples) on the GPU -> (nclips, 1, out_frames, 40) f32 MFCC:
    Envs/audioLoader.py:147-157 (torchaudio branch) + :241-252 (truncate / zero-pad).  `dataset` selects the STFT
    parameters the reference uses for clips of that dataset (audioLoader.param_dict); default = GoogleCommand's."""
    if dataset is not None:
        if dataset not in DATASET_STFT:
            raise VarHipError(f"unknown dataset {dataset!r}: one of {sorted(DATASET_STFT)}")
        n_fft, win_length, hop_length = DATASET_STFT[dataset]
    _require_cuda(pcm)
    if pcm.dtype != torch.int16 or pcm.dim() != 2:
        raise VarHipError("mfcc expects an int16 (nclips, nsamples) tensor")
    pcm = pcm.contiguous()
    stride = pcm.shape[1]
    n = pcm.shape[0] if clip_index is None else clip_index.numel()
    if clip_index is not None:
        clip_index = clip_index.to(device=pcm.device, dtype=torch.int32).contiguous()
    if lens is None:
        lens = torch.full((n,), stride, dtype=torch.int32, device=pcm.device)
    lens = lens.to(device=pcm.device, dtype=torch.int32).contiguous()
    out = torch.empty((n, 1, out_frames, 40), dtype=torch.float32, device=pcm.device)
    ctx = Context.get(pcm.device.index)
    ctx.check(ctx.lib.var_mfcc_ex(ctx.handle, current_stream_handle(), ptr(pcm), ptr(lens), ptr(clip_index), n, stride,
                                  int(out_frames), int(n_fft), int(win_length), int(hop_length), ptr(out)), "var_mfcc_ex")
    return out


def mfcc_psf(pcm, lens=None, out_frames=600, clip_index=None):
    """int16 PCM (nclips, nsamples) on the GPU -> (nclips, 1, out_frames, 40) f32: the python_speech_features
    branch of Envs/audioLoader.py:158-161 (iTHOR / FSC clips, signal not normalised) + :241-252."""
    _require_cuda(pcm)
    if pcm.dtype != torch.int16 or pcm.dim() != 2 or pcm.shape[1] % 2:
        raise VarHipError("mfcc_psf expects an int16 (nclips, even nsamples) tensor")
    pcm = pcm.contiguous()
    stride = pcm.shape[1]
    n = pcm.shape[0] if clip_index is None else clip_index.numel()
    if clip_index is not None:
        clip_index = clip_index.to(device=pcm.device, dtype=torch.int32).contiguous()
    if lens is None:
        lens = torch.full((n,), stride, dtype=torch.int32, device=pcm.device)
    lens = lens.to(device=pcm.device, dtype=torch.int32).contiguous()
    out = torch.empty((n, 1, out_frames, 40), dtype=torch.float32, device=pcm.device)
    ctx = Context.get(pcm.device.index)
    ctx.check(ctx.lib.var_mfcc_psf(ctx.handle, current_stream_handle(), ptr(pcm), ptr(lens), ptr(clip_index), n, stride,
                                   int(out_frames), ptr(out)), "var_mfcc_psf")
    return out


def inbatch_contrastive_loss(anchor, cand, target, tau=0.1, inv_count=None):
    """In-batch-negatives contrastive head (csrc/inbatch.hip; an extension of the reference's triplet loss, BASELINE
    config 3): anchor (B,3), cand (M,3) = every candidate sound embedding of the global batch, target (B) int32 =
    column of each sample's positive.  Returns (loss[1], g_anchor (B,3), g_cand (M,3)); g_cand is this rank's partial."""
    _require_cuda(anchor, cand, target)
    a, cnd = anchor.contiguous().float(), cand.contiguous().float()
    t = target.to(torch.int32).contiguous()
    B, M = a.shape[0], cnd.shape[0]
    ctx = Context.get(a.device.index)
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    ga, gc = torch.empty_like(a), torch.empty_like(cnd)
    scratch = torch.empty(2 * B, dtype=torch.float32, device=a.device)
    ctx.check(ctx.lib.var_inbatch_loss_fwd_bwd(ctx.handle, current_stream_handle(), ptr(a), ptr(cnd), ptr(t), B, M, float(tau),
                                               float(1.0 / B if inv_count is None else inv_count), ptr(scratch), ptr(loss),
                                               ptr(ga), ptr(gc)), "var_inbatch_loss_fwd_bwd")
    return loss, ga, gc
