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
    with pytest.raises(NotImplementedError):
        MVTracker(corr_n_groups=3)  # (grouped dots: a power of two <= fmaps_dim / 8)


@pytest.mark.parametrize("name", ["g4_xyz", "g2_nooffset", "g1_k8_xyz"])
def test_forward_corr_options(monkeypatch, golden, name):
    """Non-default correlation layouts through the host logic (token width, input transform shape, correlation row layout) on the
    mocked kernels, against the reference fixture."""
    from test_oracle_golden import CORR_OPT_CASES, corr_opts_clip
    hip_mock.install(monkeypatch)
    g = golden("e2e_corr_opts")
    m = MVTracker(hidden_size=256, **CORR_OPT_CASES[name]).eval()
    assert m.updateformer_input_dim == int(g[name + "_token_dim"])
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    clip = corr_opts_clip(g)
    r = m(T(clip["rgbs"]), T(clip["depths"]), T(clip["query_points"]), T(clip["intrs"]), T(clip["extrs"]), iters=3)
    ref = g[name + "_traj"]
    assert np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max() < 1e-3
    np.testing.assert_allclose(r["vis_e"].numpy(), g[name + "_vis"], atol=2e-2)  # (mocked kernels: torch arithmetic, layout check)


def test_frames_too_small_for_the_pyramid_raise_a_clear_error(monkeypatch):
    """One view of 112 x 80 pixels leaves 3 x 2 = 6 points at the coarsest correlation level, fewer than K = 16 neighbours: the
    reference fails inside its kNN there (mvtracker.py:26-90: topk / knn_query with k above the number of points); here the frame
    store says what is wrong before any kernel is asked for it (found by tests/checks/fuzz_predictor.py)."""
    hip_mock.install(monkeypatch)
    m = MVTracker(hidden_size=256).eval()
    clip = synth.make_clip(1, V=1, T=8, H=112, W=80, N=2)
    with pytest.raises(ValueError, match="fewer than corr_neighbors"):
        m(T(clip["rgbs"]), T(clip["depths"]), T(clip["query_points"]), T(clip["intrs"]), T(clip["extrs"]))


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


def test_sample_io_roundtrip(tmp_path):
    """Sample NPZ (demo.py:650, 922-929) -> tensors with a batch dimension, uint8 frames kept uint8; result NPZ keys of
    demo.py:1093-1118."""
    import numpy as np
    from mvtracker_amd import sample_io
    rng = np.random.default_rng(0)
    V, Tn, H, W, N = 2, 5, 16, 24, 7
    src = dict(rgbs=rng.integers(0, 256, (V, Tn, 3, H, W), dtype=np.uint8), depths=rng.random((V, Tn, 1, H, W), dtype=np.float32) + 1,
               intrs=np.tile(np.array([[20., 0, 12], [0, 20, 8], [0, 0, 1]], np.float32), (V, Tn, 1, 1)),
               extrs=np.tile(np.eye(4, dtype=np.float32)[:3], (V, Tn, 1, 1)), query_points=rng.random((N, 4), dtype=np.float32),
               camera_ids=np.array(["a", "b"]))
    p = tmp_path / "sample.npz"
    np.savez(p, **src)
    s = sample_io.load_sample(str(p), device="cpu")
    assert s["rgbs"].dtype == torch.uint8 and tuple(s["rgbs"].shape) == (1, V, Tn, 3, H, W)
    assert tuple(s["query_points_3d"].shape) == (1, N, 4) and list(s["camera_ids"]) == ["a", "b"]
    s2 = sample_io.load_sample(str(p), device="cpu", temporal_stride=2, spatial_downsample=2)
    assert tuple(s2["rgbs"].shape) == (1, V, 3, 3, H // 2, W // 2)
    assert np.allclose(s2["intrs"][0, 0, 0].numpy(), [[10, 0, 6], [0, 10, 4], [0, 0, 1]])
    out = tmp_path / "res.npz"
    sample_io.save_result(str(out), torch.zeros(1, Tn, N, 3), torch.ones(1, Tn, N, dtype=torch.bool), s)
    z = np.load(out, allow_pickle=False)
    assert z["tracks_3d"].shape == (Tn, N, 3) and z["visibilities"].shape == (Tn, N) and z["query_points"].shape == (N, 4)
    assert z["rgbs"].dtype == np.uint8 and str(z["tracker"]) == "mvtracker"


def test_entry_points_run_under_their_tensors_device(monkeypatch):
    """launches go to the CURRENT device's stream, so every public entry point must make its tensors'
    device current.  ``hip.guarded`` does that through ``hip.device_guard``; here the guard is recorded."""
    import torch
    from mvtracker_amd import hip
    from mvtracker_amd.adapter import assign_views
    from mvtracker_amd.parallel import ShardedTracker
    from mvtracker_amd.predictor import EvaluationPredictor
    from mvtracker_amd.tracker import MVTracker
    for fn in (MVTracker.forward, MVTracker.build_frame_store, MVTracker.encode_images, MVTracker.fill_frame_features,
               MVTracker.refine_window, MVTracker.update_former, EvaluationPredictor.forward, ShardedTracker.__call__, assign_views):
        assert hasattr(fn, "__wrapped__"), fn
    seen = []

    class Ctx:
        def __enter__(self):
            return None

        def __exit__(self, *a):
            return False

    def fake_guard(t):
        seen.append(t)
        return Ctx()

    monkeypatch.setattr(hip, "device_guard", fake_guard)

    @hip.guarded
    def entry(obj, store, frame0, coords, other=None):
        return None

    x, y = torch.ones(3), torch.zeros(2)
    entry(object(), {"xyz": [y]}, 0, x, other=y)
    assert seen and seen[-1] is x  # the first tensor argument fixes the device
    entry(object(), {"xyz": [y]}, 0, None)
    assert seen[-1] is y  # no tensor argument: falls back to the tensors inside dict / list arguments
