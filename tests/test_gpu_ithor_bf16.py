"""GPU tests of the iTHOR model's bf16 mode (BASELINE config 4): the staged 11x5 sound convolution kernels of
csrc/snd_bf16.hip, layer by layer against a float64 convolution of the SAME bf16-rounded operands (so the only
difference left is the fp32 summation order: tolerance 2e-5 of the layer's scale)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(h):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 600, 40), representationDim=3)


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def sounds(B, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 1, 600, 40, generator=g) * 3).cuda(), (torch.randn(B, 1, 600, 40, generator=g) * 3).cuda()


@pytest.mark.parametrize("B", [1, 3])
def test_conv2_forward_bf16_kernel_vs_float64_on_rounded_operands(var_amd, B):
    torch.manual_seed(5)
    m = var_amd.IthorVARPretextNet(cfg(96)).to("cuda").set_precision("bf16")
    pos, neg = sounds(B, 11 + B)
    with torch.no_grad():
        m(None, pos, neg)
    from var_amd._lib import Context
    ctx = Context.get(0)
    n = 2 * B
    s1 = ctx.debug_buffer("ithor_s1")[:n * 64 * 300 * 20].view(n, 64, 300, 20).cpu()
    s2 = ctx.debug_buffer("ithor_s2")[:n * 64 * 150 * 13].view(n, 64, 150, 13).cpu()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    wk = [k for k, v in sd.items() if tuple(v.shape) == (64, 64, 11, 5)][0]
    w, b = sd[wk], sd[wk.replace("weight", "bias")]
    ref = torch.relu(torch.nn.functional.conv2d(bf16_round(s1), bf16_round(w), b.double(), stride=2, padding=(5, 5)))
    assert ref.shape == s2.shape
    scale = float(ref.abs().max())
    assert scale > 0.1
    err = float((s2.double() - ref).abs().max())
    assert err < 2e-5 * max(scale, 1.0), (err, scale)
    assert float((s2 > 0).float().mean()) > 0.05                       # not trivially all-zero
