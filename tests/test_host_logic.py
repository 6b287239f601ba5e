"""Host-side logic of mvtracker_amd (window bookkeeping, buffer layouts, weight packing, launch
order) exercised on CPU: the ctypes wrappers are monkeypatched with tests/hip_mock.py, and the
results are compared with the oracle.  The real kernels are tested by the `-m gpu` tests."""
import numpy as np
import pytest
import torch

from mvtracker_amd import synth
from mvtracker_amd.tracker import MVTracker
from oracle import mvt_oracle as O

import hip_mock

CFG = O.TrackerConfig()


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture()
def model(monkeypatch):
    hip_mock.install(monkeypatch)
    m = MVTracker(hidden_size=256).eval()
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


def test_state_dict_keys_match_reference(golden):
    m = MVTracker(hidden_size=256)
    g = golden("state_dict_shapes")
    sd = m.state_dict()
    assert sorted(sd) == list(g["keys"])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == list(g["shapes"])
    assert m.updateformer_input_dim == 581


def test_constructor_rejects_unsupported():
    with pytest.raises(NotImplementedError):
        MVTracker(normalize_scene_in_fwd_pass=True)
    with pytest.raises(NotImplementedError):
        MVTracker(corr_filter_invalid_depth=True)


def test_cpu_tensors_are_refused():
    m = MVTracker(hidden_size=256)
    clip = synth.make_clip(1, V=1, T=8, H=64, W=64, N=2)
    from mvtracker_amd import hip
    with pytest.raises(hip.HipError):
        m(T(clip["rgbs"]), T(clip["depths"]), T(clip["query_points"]), T(clip["intrs"]), T(clip["extrs"]))


def test_encoder_sequence(model, golden):
    g = golden("encoder_64x96")
    img = T(g["img"])  # already normalised: feed through the conv stack directly
    x4 = torch.zeros(2, 64, 96, 4)
    x4[..., :3] = img.permute(0, 2, 3, 1)
    out = torch.zeros(2, 16, 24, 128)
    model._encode(model._pack(torch.device("cpu")), x4, 2, 64, 96, out)
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).numpy(), g["out"], rtol=1e-3, atol=2e-4)


def test_updateformer_sequence(model, golden):
    g = golden("updateformer_16x12")
    out = model.update_former(T(g["x"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("fused", [True, False])
def test_updateformer_sequence_bf16(model, golden, fused):
    """bf16 paths: fused block kernels (5 launches per layer) and the unfused sequence must agree with the fp32 golden
    output to bf16 accuracy -- a buffer-reuse mistake in the sequencing would be off by O(1)."""
    g = golden("updateformer_16x12")
    model.precision = "bf16"
    model.fuse_blocks = fused
    out = model.update_former(T(g["x"]))
    ref = g["out"]
    err = np.abs(out.numpy() - ref).max() / np.abs(ref).max()
    assert err < 3e-2, err


@pytest.mark.parametrize("name", ["e2e_tiny", "e2e_two_windows", "e2e_short_clip"])
def test_forward_sequence(model, golden, name):
    g = golden(name)
    kw = dict(seed=int(g["seed"]), V=int(g["V"]), T=int(g["T"]), H=int(g["H"]), W=int(g["W"]), N=int(g["N"]))
    if "late_queries" in g.files:
        kw.update(late_queries=bool(g["late_queries"]), query_frames=tuple(int(x) for x in g["query_frames"]))
    clip = synth.make_clip(**kw)
    r = model(T(clip["rgbs"]), T(clip["depths"]), T(clip["query_points"]), T(clip["intrs"]), T(clip["extrs"]), iters=4)
    assert len(model.last_windows) == int(g["n_windows"])
    ref = g["traj_exact"]
    assert r["traj_e"].shape == ref.shape
    rel = np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max()
    assert rel < 1e-3, rel
    np.testing.assert_allclose(r["vis_e"].numpy(), g["vis_exact"], atol=5e-3)
    np.testing.assert_allclose(r["feat_init"].numpy(), g["feat_init_exact"], rtol=1e-3, atol=1e-4)
    model.check_finite()
