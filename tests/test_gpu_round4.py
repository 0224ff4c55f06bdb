"""GPU tests added in round 4 (run with -m gpu on an MI355X): the device-side hand-over between the two streams of a training
step."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(h=84):
    return types.SimpleNamespace(img_dim=(3, h, h), sound_dim=(1, 100, 40), representationDim=3)


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def make_model(var_amd, sd, h=84):
    m = var_amd.VARPretextNet(cfg(h))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to("cuda")


@pytest.mark.parametrize("B,h", [(48, 84), (256, 84), (256, 96)])
def test_device_side_stream_join_equals_the_graph_edge(var_amd, golden_dir, B, h):
    """Round 4: inside a training step no graph edge crosses the two streams between the fork and the final join -- a barrier
    packet in front of the first backward kernel cost ~10 us of every replayed step.  The image rows of the heads' backward need
    the sound embeddings' partials, the sound rows the image's: the last workgroup of each producer (sound heads' forward / the
    conv 3-5 kernel) counts a flag up and does not end before the other stream's flag has been counted up too, and the rows read
    the other stream's partials with agent-scope loads (csrc/var_common.h: join_signal; csrc/heads.hip).  Same arithmetic either
    way: replayed steps with the flags (default) and with the edges (var_set_streams bit 6) from the same start must leave
    bit-identical losses and parameters, eager steps (which always keep the edges) likewise, and no wait may have timed out
    (var_join_status).  (256, 96): the replayed step on the 96 x 96 forms of the image kernels."""
    from var_amd._lib import Context
    sd = load(golden_dir, "kuka_weights.npz")
    ctx = Context.get(0)
    pool = var_amd.SyntheticTripletPool(4 * B, hw=h, seed=5, clips_per_class=4).freeze_pairs()
    table = pool.index_table(B, 4)[:4].contiguous()
    before = ctx.join_timeouts()
    out = {}
    for mode, mask in (("flag", 3), ("edge", 3 | 64)):
        old = ctx.set_streams(mask)
        try:
            m = make_model(var_amd, sd, h)
            tr = var_amd.VARTrainer(m, lr=1e-3, weight_decay=1e-6)
            replay, _ = tr.capture_epoch_steps(pool.images, pool.clips, B, table)
            losses = [float(replay().item()) for _ in range(6)]
            torch.cuda.synchronize()
            r = table[2]
            losses.append(float(tr.step_from_dataset(pool.images, r[:B], pool.clips, r[B:3 * B], r[3 * B:]).item()))   # an eager step too
            torch.cuda.synchronize()
            out[mode] = (losses, m.flat_parameters().cpu().numpy().copy())
            del tr, m
        finally:
            ctx.set_streams(old)
    assert ctx.join_timeouts() == before, "a device-side wait for the sound branch gave up"
    assert out["flag"][0] == out["edge"][0], (out["flag"][0], out["edge"][0])
    assert np.array_equal(out["flag"][1], out["edge"][1])
    assert all(np.isfinite(out["flag"][0]))
