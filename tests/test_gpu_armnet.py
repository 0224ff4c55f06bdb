"""GPU parity of the RL actor-critic forward (var_armnet_forward through
ArmNetPolicy.act) against the fixture the reference's Policy produced (tests/golden/armnet_b8.npz) and against the CPU
oracle with identical weights."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.torch_oracle import armnet_seeded  # noqa: E402  (checker only)


class Box:
    def __init__(self, n):
        self.shape = (n,)


CFG = types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2)
KW = {'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512, 'actionHiddenSize': 128}


@pytest.fixture(scope="module")
def var_amd():
    import var_amd as m
    assert torch.cuda.is_available()
    return m


@pytest.fixture(scope="module")
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "armnet_b8.npz")))


def cuda(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def make(var_amd, ref):
    m = var_amd.ArmNetPolicy(None, Box(2), config=CFG, base='arm_VAR', base_kwargs=KW)
    m.load_state_dict(ref.state_dict())                       # same 63 keys and shapes as the reference's Policy
    return m.to("cuda")


def obs_of(fx, u8):
    img = cuda(fx['image'])
    return {'image': img if u8 else (img / 255.).float(), 'image_feat': cuda(fx['image_feat']),
            'robot_pose': cuda(fx['robot_pose']), 'goal_sound_feat': cuda(fx['goal_sound_feat'])}


def test_act_vs_reference_fixture_and_oracle(var_amd, fx):
    ref = armnet_seeded(int(fx["seed"]))
    m = make(var_amd, ref)
    assert [k for k, _ in m.state_dict().items()] == [str(k) for k in fx["names"]]
    assert m.is_recurrent and m.recurrent_hidden_state_size == 512
    for u8 in (False, True):
        v, a, lp, h = m.act(obs_of(fx, u8), cuda(fx['rnn_hxs']), cuda(fx['masks']), deterministic=True)
        v2, a2, _, h2 = m.act(obs_of(fx, u8), h, torch.ones(8, 1, device="cuda"), deterministic=True)
        for got, name in ((v, 'value'), (a, 'action'), (lp, 'action_log_probs'), (h, 'rnn_hxs_out'), (v2, 'value2'),
                          (a2, 'action2'), (h2, 'rnn_hxs_out2')):
            np.testing.assert_allclose(got.cpu().numpy(), fx[name], rtol=1e-3, atol=1e-4, err_msg=name)
    # identical weights on both sides: tighter
    obs_cpu = {'image': (torch.from_numpy(fx['image']) / 255.).float(), 'image_feat': torch.from_numpy(fx['image_feat']),
               'robot_pose': torch.from_numpy(fx['robot_pose']), 'goal_sound_feat': torch.from_numpy(fx['goal_sound_feat'])}
    with torch.no_grad():
        rv, ra, rlp, rh, rf = ref.act_deterministic(obs_cpu, torch.from_numpy(fx['rnn_hxs']), torch.from_numpy(fx['masks']))
    v, a, lp, h = m.act(obs_of(fx, False), cuda(fx['rnn_hxs']), cuda(fx['masks']), deterministic=True)
    np.testing.assert_allclose(v.cpu().numpy(), rv.numpy(), atol=2e-5)
    np.testing.assert_allclose(a.cpu().numpy(), ra.numpy(), atol=2e-5)
    np.testing.assert_allclose(h.cpu().numpy(), rh.numpy(), atol=2e-5)
    np.testing.assert_allclose(m.get_value(obs_of(fx, False), cuda(fx['rnn_hxs']), cuda(fx['masks'])).cpu().numpy(), rv.numpy(), atol=2e-5)


def test_sampling_and_rejections(var_amd, fx):
    ref = armnet_seeded(int(fx["seed"]))
    m = make(var_amd, ref)
    torch.manual_seed(0)
    v, a, lp, h = m.act(obs_of(fx, True), cuda(fx['rnn_hxs']), cuda(fx['masks']))
    assert a.shape == (8, 2) and lp.shape == (8, 1) and torch.isfinite(a).all()
    mean = m.act(obs_of(fx, True), cuda(fx['rnn_hxs']), cuda(fx['masks']), deterministic=True)[1]
    # log-prob of the sampled action under N(mean, 1) (logstd is zero at initialisation)
    expect = (-0.5 * (a - mean) ** 2 - 0.5 * np.log(2 * np.pi)).sum(-1, keepdim=True)
    np.testing.assert_allclose(lp.cpu().numpy(), expect.cpu().numpy(), atol=1e-5)
    with pytest.raises(NotImplementedError):
        m.evaluate_actions(None, None, None, None)
    with pytest.raises(var_amd.VarHipError):
        m.act({k: t.cpu() for k, t in obs_of(fx, True).items()}, cuda(fx['rnn_hxs']), cuda(fx['masks']))
    with pytest.raises(var_amd.VarHipError):
        var_amd.ArmNetPolicy(None, Box(2), config=types.SimpleNamespace(img_dim=(3, 84, 84), representationDim=3, robotStateDim=2),
                             base='arm_VAR', base_kwargs=KW)


@pytest.mark.parametrize("B", [1, 5, 8])
def test_fused_small_batch_chain_equals_the_per_layer_path(var_amd, fx, B):
    """B <= 8 (the RL stage's envs) takes the one-launch MLP chain after the convolutions (csrc/armnet.hip:
    armnet_chain_kernel); larger batches take one launch per layer.  The same rows through both: a batch of B alone, and
    as the first B rows of a batch of 12.  Two consecutive steps (the second from the first's hidden state)."""
    ref = armnet_seeded(int(fx["seed"]))
    m = make(var_amd, ref)
    g = torch.Generator().manual_seed(5)
    big = {'image': torch.randint(0, 256, (12, 3, 96, 96), dtype=torch.uint8, generator=g).cuda(),
           'image_feat': torch.randn(12, 3, generator=g).cuda(), 'robot_pose': torch.randn(12, 2, generator=g).cuda(),
           'goal_sound_feat': torch.randn(12, 3, generator=g).cuda()}
    hxs = torch.randn(12, 512, generator=g).cuda() * 0.3
    masks = torch.tensor([[1.], [0.], [1.], [1.], [0.], [1.], [1.], [1.], [1.], [0.], [1.], [1.]]).cuda()
    small = {k: v[:B].contiguous() for k, v in big.items()}
    v1, a1, _, h1 = m.act(small, hxs[:B].contiguous(), masks[:B].contiguous(), deterministic=True)
    v2, a2, _, h2 = m.act(big, hxs, masks, deterministic=True)
    for got, want in ((v1, v2), (a1, a2), (h1, h2)):
        np.testing.assert_allclose(got.cpu().numpy(), want[:B].cpu().numpy(), rtol=0, atol=2e-5)
    v3, a3, _, h3 = m.act(small, h1, torch.ones(B, 1, device="cuda"), deterministic=True)
    v4, a4, _, h4 = m.act(big, h2, torch.ones(12, 1, device="cuda"), deterministic=True)
    for got, want in ((v3, v4), (a3, a4), (h3, h4)):
        np.testing.assert_allclose(got.cpu().numpy(), want[:B].cpu().numpy(), rtol=0, atol=5e-5)


def test_band_convolutions_equal_the_gather_gemm_path(var_amd, fx):
    """Up to 64 images the image stack runs on the LDS-band kernels of csrc/c3f.h (conv 1 + filter pack, conv 2..6 with the
    pools fused, conv 7/8 with K split over a workgroup's waves); larger batches keep the gather-GEMM + split-K finish + pool
    launches.  The same 8 rows through both: alone (band kernels, fused chain) and as the first rows of a batch of 66."""
    ref = armnet_seeded(int(fx["seed"]))
    m = make(var_amd, ref)
    g = torch.Generator().manual_seed(11)
    n = 66
    big = {'image': torch.randint(0, 256, (n, 3, 96, 96), dtype=torch.uint8, generator=g).cuda(),
           'image_feat': torch.randn(n, 3, generator=g).cuda(), 'robot_pose': torch.randn(n, 2, generator=g).cuda(),
           'goal_sound_feat': torch.randn(n, 3, generator=g).cuda()}
    hxs = torch.randn(n, 512, generator=g).cuda() * 0.3
    masks = (torch.rand(n, 1, generator=g) > 0.2).float().cuda()
    for B in (8, 40):                                      # 40: band kernels + per-layer MLP
        small = {k: v[:B].contiguous() for k, v in big.items()}
        v1, a1, _, h1 = m.act(small, hxs[:B].contiguous(), masks[:B].contiguous(), deterministic=True)
        v2, a2, _, h2 = m.act(big, hxs, masks, deterministic=True)
        for got, want in ((v1, v2), (a1, a2), (h1, h2)):
            np.testing.assert_allclose(got.cpu().numpy(), want[:B].cpu().numpy(), rtol=0, atol=2e-5)
    # float images take the same kernels' other instantiation
    fl = {k: (v[:8].float() / 255. if k == 'image' else v[:8].contiguous()) for k, v in big.items()}
    v3, a3, _, h3 = m.act(fl, hxs[:8].contiguous(), masks[:8].contiguous(), deterministic=True)
    v4, a4, _, h4 = m.act({k: v[:8].contiguous() for k, v in big.items()}, hxs[:8].contiguous(), masks[:8].contiguous(), deterministic=True)
    for got, want in ((v3, v4), (a3, a4), (h3, h4)):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-5)


def test_replayed_forward_is_stable_and_sees_new_inputs_and_weights(var_amd, fx):
    """The chain's workgroups hand vectors over as (value, epoch) pairs and poll their inputs: the epoch lives on the device
    because a captured graph replays with frozen arguments.  200 replays of one captured forward must return the first replay's
    bits every time (a stale pair accepted once would show), follow the static input buffers when they change, and follow the
    parameters when they change in place (the filters are re-packed inside every forward)."""
    ref = armnet_seeded(int(fx["seed"]))
    m = make(var_amd, ref)
    obs = {k: v.clone() for k, v in obs_of(fx, True).items()}
    hxs, masks = cuda(fx['rnn_hxs']).clone(), cuda(fx['masks']).clone()
    eager = [t.clone() for t in m._base_forward(obs, hxs, masks)]
    g = var_amd._lib.new_graph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m._base_forward(obs, hxs, masks)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            outs = m._base_forward(obs, hxs, masks)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    first = [t.clone() for t in outs]
    for a, b in zip(first, eager):
        assert torch.equal(a, b)
    for i in range(200):
        g.replay()
        if i % 50 == 49:
            torch.cuda.synchronize()
            for a, b in zip(outs, first):
                assert torch.equal(a, b), f"replay {i}"
    # new inputs through the same static buffers
    obs['robot_pose'].add_(0.25)
    obs['image'].copy_(torch.flip(obs['image'], dims=[3]))
    want = [t.clone() for t in m._base_forward(obs, hxs, masks)]
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, want):
        assert torch.equal(a, b)
    assert not torch.equal(outs[0], first[0])
    # parameters changed in place (a PPO update between two acting steps)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)
    want = [t.clone() for t in m._base_forward(obs, hxs, masks)]
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, want):
        assert torch.equal(a, b)


def test_chain_time_out_is_reported_leaves_nothing_behind_and_clears(var_amd, fx):
    """The reference's Policy.act (models/ppo/model.py:57-69) cannot fail; the fused small-batch chain can -- it needs its 128
    workgroups resident at once.  Fault injection (var_debug_armnet_drop_workgroup: the next chain launch runs one
    workgroup short, so the vectors it owes never arrive): the launch ends, its outputs are NaN, var_armnet_status reads 1;
    the NEXT launch is bit-identical to an undisturbed one and the status says "an earlier launch"; the record clears."""
    import ctypes
    from var_amd._lib import Context
    ref = armnet_seeded(int(fx["seed"]))
    m = make(var_amd, ref)
    obs, hxs, masks = obs_of(fx, True), cuda(fx['rnn_hxs']), cuda(fx['masks'])
    good = [t.clone() for t in m.act(obs, hxs, masks, deterministic=True)]
    torch.cuda.synchronize()
    assert m.chain_status() == 0
    ctx = Context.get(0)
    assert ctx.lib.var_debug_armnet_drop_workgroup(ctx.handle) == 0
    with torch.no_grad():
        bad = m._base_forward(obs, hxs, masks)                    # (act() would raise in torch.distributions: the mean is NaN)
    torch.cuda.synchronize()
    assert torch.isnan(bad[0]).all() and torch.isnan(bad[2]).all(), "a timed-out chain must not return half-updated values"
    assert m.chain_status() == 1
    again = m.act(obs, hxs, masks, deterministic=True)
    torch.cuda.synchronize()
    for g, a in zip(good, again):
        assert torch.equal(g, a)
    assert m.chain_status() == 0x40000001
    m.clear_chain_status()
    assert m.chain_status() == 0
    # in-place state update is refused (the chain reads rnn_hxs from every workgroup of its GRU stage while one writes the new state)
    word = ctypes.c_uint(0)
    assert ctx.lib.var_armnet_status(ctx.handle, ctypes.byref(word)) == 0 and word.value == 0
    flat = m._flat
    img = obs['image'].reshape(8, -1, 96, 96).contiguous()
    f = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731
    out = [torch.empty(8, n, device="cuda") for n in (1, 128, 2)]
    h = hxs.clone().float().contiguous()
    rc = ctx.lib.var_armnet_forward(ctx.handle, None, f(flat), f(img), 1, img.stride(0), f(obs['image_feat'].float().contiguous()),
                                    f(obs['robot_pose'].float().contiguous()), f(obs['goal_sound_feat'].float().contiguous()),
                                    f(h), f(masks.float().contiguous()), 8, f(out[0]), f(out[1]), f(out[2]), f(h))
    assert rc != 0
