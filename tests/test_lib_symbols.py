"""The C-ABI library loads on a CPU-only host and exports every symbol include/var_hip.h declares
(no compute is called here); host-side rules of the data layer."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    import var_amd
    hdr = open(os.path.join(ROOT, "include", "var_hip.h")).read()
    declared = set(re.findall(r"\b(var_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = ctypes.CDLL(var_amd.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in var_hip.h but not exported"
    assert lib.var_param_count() == var_amd.N_PARAMS == 213478
    from var_amd._lib import EXPORTED_SYMBOLS
    assert set(EXPORTED_SYMBOLS) <= declared


def test_no_gpu_means_loud_failure():
    import types
    import var_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
    m = var_amd.VARPretextNet(cfg)
    with pytest.raises(var_amd.VarHipError):
        m(torch.zeros(1, 3, 84, 84), None, None)
    with pytest.raises(var_amd.VarHipError):
        var_amd.VARTrainer(m)
    with pytest.raises(var_amd.VarHipError):
        var_amd.VARPretextNet(types.SimpleNamespace(img_dim=(3, 64, 64), sound_dim=(1, 100, 40), representationDim=3))


def test_state_dict_layout_and_seeded_init_match_reference(golden_dir):
    import types
    import var_amd
    cfg = types.SimpleNamespace(img_dim=(3, 84, 84), sound_dim=(1, 100, 40), representationDim=3)
    torch.manual_seed(453)
    m = var_amd.VARPretextNet(cfg)
    sd = dict(np.load(os.path.join(golden_dir, "kuka_weights.npz")))
    got = m.state_dict()
    assert list(got.keys()) == list(sd.keys())
    for k in sd:                                     # same RNG draws as the reference constructor
        assert np.array_equal(got[k].numpy(), sd[k]), k
    m.load_state_dict({k: torch.from_numpy(v * 2) for k, v in sd.items()})
    flat = m.flat_parameters().numpy()
    assert np.array_equal(flat[:864], (sd['imgBranch.0.weight'] * 2).reshape(-1))   # params alias the arena


def test_negative_id_rule_and_padding():
    import var_amd
    # dataset.py:73-78: collision with the label maps to taskNum (the "empty" class)
    assert var_amd.choose_negative_id(2, 4, rand_int=lambda lo, hi: 2) == 4
    assert var_amd.choose_negative_id(2, 4, rand_int=lambda lo, hi: 1) == 1
    assert var_amd.choose_negative_id(2, 4, stored=3) == 3
    f = np.ones((51, 40), np.float32)
    p = var_amd.process_sound_feat(f)
    assert p.shape == (1, 100, 40) and p[0, :51].all() and not p[0, 51:].any()
    assert var_amd.process_sound_feat(np.ones((101, 40), np.float32)).shape == (1, 100, 40)
    assert abs(var_amd.multistep_lr(1e-4, [10, 30, 50], 0.2, 30) - 4e-6) < 1e-18


def test_ithor_and_actor_critic_module_layouts_match_reference_fixtures(golden_dir):
    """CPU-side: the drop-in modules carry the reference's state_dict keys/shapes, the seed reproduces the iTHOR
    weights bit for bit, the arena sizes equal the C ABI's counts, and CPU use fails loudly (no fallback)."""
    import types
    import var_amd
    lib = var_amd.load_library()
    g = np.load(os.path.join(golden_dir, "ithor_h96.npz"))
    torch.manual_seed(int(g["seed"]))
    m = var_amd.IthorVARPretextNet(types.SimpleNamespace(img_dim=(3, 96, 96), sound_dim=(1, 600, 40), representationDim=3))
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["names"]]
    for k, v in sd.items():
        f = v.numpy().reshape(-1).astype(np.float64)
        assert tuple(v.shape) == tuple(g["shape." + k])
        assert np.array_equal(np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]]), g["check." + k]), k
    assert m.flat_parameters().numel() == lib.var_ithor_param_count() == 3849126
    with pytest.raises(var_amd.VarHipError):
        m(torch.zeros(1, 3, 96, 96), None, None)
    with pytest.raises(var_amd.VarHipError):
        var_amd.IthorTrainer(m)

    class Box:
        shape = (2,)
    a = np.load(os.path.join(golden_dir, "armnet_b8.npz"))
    ac = var_amd.ArmNetPolicy(None, Box(), config=types.SimpleNamespace(img_dim=(3, 96, 96), representationDim=3, robotStateDim=2),
                              base='arm_VAR', base_kwargs={'recurrent': True, 'recurrentInputSize': 128, 'recurrentSize': 512,
                                                           'actionHiddenSize': 128})
    sd = ac.state_dict()
    assert list(sd.keys()) == [str(k) for k in a["names"]]
    assert all(tuple(v.shape) == tuple(a["shape." + k]) for k, v in sd.items())
    assert sum(v.numel() for v in sd.values()) == lib.var_armnet_param_count()
    with pytest.raises(var_amd.VarHipError):
        ac.act({'image': torch.zeros(1, 3, 96, 96)}, torch.zeros(1, 512), torch.ones(1, 1))
