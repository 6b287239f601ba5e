"""Pin the CPU oracle (oracle/mvt_oracle.py) against golden vectors produced by the reference.

The .npz files were written by tests/golden/make_golden.py, which imports the reference from
/root/reference in the build container.  These tests need neither the reference nor a GPU.
"""
import numpy as np
import pytest
import torch

from mvtracker_amd import synth
from oracle import mvt_oracle as O

CFG = O.TrackerConfig()


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def W():
    return O.make_weights(CFG, seed=0)


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_state_dict_contract(golden):
    g = golden("state_dict_shapes")
    shapes = O.state_dict_shapes(CFG)
    assert sorted(shapes) == list(g["keys"])
    assert [str(tuple(shapes[k])) for k in sorted(shapes)] == list(g["shapes"])
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"]) == 22607356
    assert CFG.token_dim == int(g["token_dim"]) == 581


def test_weights_recipe_matches_synth(W):
    sd = synth.make_state_dict(O.state_dict_shapes(CFG), seed=0)
    for k, v in W.items():
        assert np.array_equal(v.numpy(), sd[k]), k


def test_encoder(golden, W):
    g = golden("encoder_64x96")
    close(O.encoder(W, T(g["img"])), g["out"], rtol=1e-4, atol=1e-5)


def test_pyramid(golden):
    g = golden("pyramid_small")
    for lvl in range(3):
        xyz, fvec, valid = O.pointcloud_level(T(g["fmaps"]), T(g["depths_strided"]), T(g["intrs"]), T(g["extrs"]), 4, lvl,
                                              return_valid=True)
        close(xyz, g[f"xyz{lvl}"])
        assert np.array_equal(fvec.numpy(), g[f"fvec{lvl}"])
        assert np.array_equal(valid.numpy(), g[f"valid{lvl}"])


@pytest.mark.parametrize("mode", ["cdist", "exact"])
def test_corr_sample(golden, mode):
    g = golden("corr_sample_small")
    out, idx = O.corr_sample(T(g["xyz"]), T(g["fvec"]), T(g["targets"]), T(g["coords"]), 16, 1, True, False, mode,
                             return_idx=True)
    assert np.array_equal(idx.numpy(), g["idx_" + mode])  # integer indices bit-exact
    close(out, g["out_" + mode])
    d2, _ = O.knn(16, T(g["xyz"]), T(g["coords"]), mode)
    ref_d = g["dist_" + mode]
    close(np.sqrt(d2.numpy()) if mode == "exact" else d2, ref_d, rtol=1e-6)


def test_corr_sample_groups_xyz(golden):
    g = golden("corr_sample_small")
    out = O.corr_sample(T(g["xyz"]), T(g["fvec"]), T(g["targets"]), T(g["coords"]), 8, 4, True, True, "exact")
    close(out, g["out_exact_k8_g4_xyz"])


@pytest.mark.parametrize("r", [3, 4])
def test_window_corr(golden, r):
    g = golden("window_corr_small")
    pyr = O.window_corr_pyramid(T(g["fmaps"]), 3)
    close(O.window_corr_sample(pyr, T(g["targets"]), T(g["coords"]), r), g[f"out_r{r}"])


def test_embeddings(golden):
    g = golden("embeddings")
    pe = O.pos_embed_3d(582, T(g["coords0"]).reshape(-1, 3))
    assert np.array_equal(pe.numpy(), g["pos_embed"].reshape(-1, 582))  # float64, bit-exact
    te = O.sincos_1d(582, (torch.linspace(0, 11, 12).reshape(12, 1) / 12).numpy())
    assert np.array_equal(te, g["times_embed"])
    close(O.flow_embedding(T(g["flows"]), 64), g["flow_embed"], rtol=0, atol=0)


def test_updateformer(golden, W):
    g = golden("updateformer_16x12")
    close(O.update_former(W, T(g["x"]), CFG), g["out"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("mode", ["cdist", "exact"])
def test_refine_window(golden, W, mode):
    g = golden("refine_window_small")
    clip = synth.make_clip(int(g["clip_seed"]), V=2, T=12, H=128, W=128, N=12)
    fm = O.encoder(W, 2 * (T(clip["rgbs"]).reshape(-1, 3, 128, 128) / 255.0) - 1).reshape(1, 2, 12, 128, 32, 32)
    d = torch.nn.functional.interpolate(T(clip["depths"]).reshape(-1, 1, 128, 128), scale_factor=0.25, mode="nearest")
    q = T(clip["query_points"])
    n = q.shape[1]
    preds, vis = O.refine_window(W, CFG, fm, d.reshape(1, 2, 12, 1, 32, 32), T(clip["intrs"]), T(clip["extrs"]),
                                 q[:, None, :, 1:].repeat(1, 12, 1, 1), torch.full((1, 12, n, 1), 10.0),
                                 torch.ones(1, 12, n, 1, dtype=torch.bool), T(g["feat_init"]), iters=3, knn_mode=mode)
    close(torch.stack(preds), g["coords_" + mode], rtol=1e-4, atol=1e-5)
    close(vis, g["vis_" + mode], rtol=0, atol=1e-3)


@pytest.mark.parametrize("name", ["e2e_tiny", "e2e_two_windows", "e2e_short_clip"])
@pytest.mark.parametrize("mode", ["cdist", "exact"])
def test_end_to_end(golden, W, name, mode):
    g = golden(name)
    kw = dict(seed=int(g["seed"]), V=int(g["V"]), T=int(g["T"]), H=int(g["H"]), W=int(g["W"]), N=int(g["N"]))
    if "late_queries" in g.files:
        kw.update(late_queries=bool(g["late_queries"]), query_frames=tuple(int(x) for x in g["query_frames"]))
    clip = synth.make_clip(**kw)
    r = O.tracker_forward(W, CFG, T(clip["rgbs"]), T(clip["depths"]), T(clip["query_points"]), T(clip["intrs"]),
                          T(clip["extrs"]), iters=4, knn_mode=mode)
    assert len(r["windows"]) == int(g["n_windows"])
    ref = g["traj_" + mode]
    rel = np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max()
    assert rel < 1e-4, rel  # north-star tolerance: tracks within 1e-4 relative
    close(r["vis_e"], g["vis_" + mode], rtol=0, atol=1e-3)
    close(r["feat_init"], g["feat_init_" + mode], rtol=1e-4, atol=1e-5)
    inv, srt = r["inv_sort_inds"], r["sort_inds"]
    assert torch.equal(inv[srt], torch.arange(len(srt)))  # appendix D: inverse-permutation property


def test_predictor(golden, W):
    g = golden("predictor_small")
    clip = synth.make_clip(int(g["clip_seed"]), V=2, T=12, H=160, W=192, N=5)
    args = [T(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    rg, dp, intr, support = O.predictor_prepare(*args, interp_shape=(128, 160), grid_size=3, n_grids_per_view=2)
    q = torch.cat([args[2], support], 1)
    close(q, g["model_query_points"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(q[0, :, 0].long().numpy(), g["model_query_points"][0, :, 0].astype(np.int64))
    close(intr, g["model_intrs"])
    assert np.array_equal(dp[0, :, :, 0, ::8, ::8].numpy(), g["depths_sample"])
    assert np.array_equal(rg[0, :, :, :, ::8, ::8].numpy(), g["rgbs_sample"])
    r = O.predictor_forward(W, CFG, *args, interp_shape=(128, 160), grid_size=3, n_grids_per_view=2, n_iters=2)
    ref = g["traj_e"]
    assert np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max() < 1e-4
    close(r["vis_e_as_prob"], g["vis_e_as_prob"], rtol=0, atol=1e-3)
    assert r["vis_e"].dtype == torch.bool


def test_helpers(golden):
    g = golden("helpers")
    pix, z = O.project_to_view(T(g["world"]), T(g["intrs"]), T(g["extrs"]))
    close(pix, g["pix"])
    close(z, g["z"])
    close(O.bilinear_sample2d(T(g["im"]), T(g["xs"]), T(g["ys"])), g["bil"])
    close(O.grid_points(5, (48, 64)), g["grid5"])
    close(O.grid_points(3, (50, 50), center=(20.5, 31.25)), g["grid3c"])


def test_adapter_view_assignment(golden):
    """MonocularToMultiViewAdapter's integer best-view index (monocular_baselines.py:630-680), reference-generated fixture
    (tests/golden/make_golden_adapter.py): bit-exact."""
    g = golden("adapter_view_assignment")
    clip = synth.make_clip(int(g["seed"]), V=int(g["V"]), T=int(g["T"]), H=int(g["H"]), W=int(g["W"]), N=int(g["N"]), late_queries=True,
                           query_frames=(2, 5))
    bv = O.adapter_best_view(T(clip["depths"][0]), T(g["query_points"][0]), T(clip["intrs"][0]), T(clip["extrs"][0]))
    assert np.array_equal(bv.numpy(), g["best_view"])


# ---------------------------------------------------------------------------------- round-2 fixtures (make_golden_r2.py)
AC = lambda: torch.autocast("cpu", dtype=torch.bfloat16)  # noqa: E731


def test_pyramid_bf16_autocast(golden):
    """H2 quirk: under autocast the reference's point cloud itself is bf16 (pixel grid cast to the fmap dtype,
    model_utils.py:467, then two bf16 einsums) -- the oracle under autocast reproduces it bit-for-bit."""
    g = golden("pyramid_small_bf16")
    fm = T(g["fmaps"]).bfloat16()
    with AC():
        for lvl in range(3):
            xyz, fvec = O.pointcloud_level(fm, T(g["depths_strided"]), T(g["intrs"]), T(g["extrs"]), 4, lvl)
            assert xyz.dtype == torch.bfloat16 and fvec.dtype == torch.bfloat16
            assert np.array_equal(xyz.float().numpy(), g[f"xyz{lvl}"])
            assert np.array_equal(fvec.float().numpy(), g[f"fvec{lvl}"])


@pytest.mark.parametrize("mode", ["cdist", "exact"])
def test_refine_window_bf16_autocast(golden, W, mode):
    g = golden("refine_window_small_bf16")
    clip = synth.make_clip(int(g["clip_seed"]), V=2, T=12, H=128, W=128, N=12)
    with torch.no_grad(), AC():
        fm = O.encoder(W, 2 * (T(clip["rgbs"]).reshape(-1, 3, 128, 128) / 255.0) - 1).reshape(1, 2, 12, 128, 32, 32)
        d = torch.nn.functional.interpolate(T(clip["depths"]).reshape(-1, 1, 128, 128), scale_factor=0.25, mode="nearest")
        q = T(clip["query_points"])
        n = q.shape[1]
        preds, vis = O.refine_window(W, CFG, fm, d.reshape(1, 2, 12, 1, 32, 32), T(clip["intrs"]), T(clip["extrs"]),
                                     q[:, None, :, 1:].repeat(1, 12, 1, 1), torch.full((1, 12, n, 1), 10.0),
                                     torch.ones(1, 12, n, 1, dtype=torch.bool), T(g["feat_init"]).bfloat16(), iters=3, knn_mode=mode)
    if mode == "cdist":  # the very same torch ops as the reference: bit-identical under autocast too
        assert np.array_equal(torch.stack(preds).float().numpy(), g["coords_cdist"])
        assert np.array_equal(vis.float().numpy(), g["vis_cdist"])
    else:
        close(torch.stack(preds).float(), g["coords_exact"], rtol=1e-4, atol=1e-5)
        close(vis.float(), g["vis_exact"], rtol=0, atol=1e-3)


@pytest.mark.parametrize("name", ["e2e_tiny_bf16", "e2e_two_windows_bf16"])
@pytest.mark.parametrize("mode", ["cdist", "exact"])
def test_end_to_end_bf16_autocast(golden, W, name, mode):
    """the oracle under CPU bf16 autocast against the REFERENCE under CPU bf16 autocast."""
    g = golden(name)
    kw = dict(seed=int(g["seed"]), V=int(g["V"]), T=int(g["T"]), H=int(g["H"]), W=int(g["W"]), N=int(g["N"]))
    if "late_queries" in g.files:
        kw.update(late_queries=bool(g["late_queries"]), query_frames=tuple(int(x) for x in g["query_frames"]))
    clip = synth.make_clip(**kw)
    a = [T(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    with torch.no_grad():
        with AC():
            r = O.tracker_forward(W, CFG, *a, iters=4, knn_mode=mode)
        r32 = O.tracker_forward(W, CFG, *a, iters=4, knn_mode=mode)
    ref = g["traj_" + mode]
    rel = np.abs(r["traj_e"].float().numpy() - ref).max() / np.abs(ref).max()
    verr = np.abs(r["vis_e"].float().numpy() - g["vis_" + mode]).max()
    print(f"{name}/{mode}: oracle-autocast vs reference-autocast tracks {rel:.2e} vis {verr:.2e}")
    if mode == "cdist":
        assert rel == 0.0 and verr == 0.0  # same ops, same rounding points
    else:
        assert rel < 1e-4 and verr < 1e-3
    # and the fp32 run recorded beside it (the distance autocast moves the reference itself)
    assert np.abs(r32["traj_e"].numpy() - g["traj_fp32_" + mode]).max() / np.abs(ref).max() < 1e-4


def test_predictor_single_point_local_grids(golden, W):
    """single_point mode with local support grids (evaluation_predictor_3dpt.py:191-277): the per-query model inputs and
    the kept tracks against the reference."""
    g = golden("predictor_single_point")
    clip = synth.make_clip(int(g["clip_seed"]), V=2, T=12, H=128, W=128, N=3)
    args = [T(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    args[2] = T(g["query_points"])
    r = O.predictor_forward_single_point(W, CFG, *args, interp_shape=None, grid_size=2, local_grid_size=3, local_extent=20,
                                         n_iters=2)
    assert len(r["per_query_inputs"]) == int(g["n_calls"]) == 3
    for i, q in enumerate(r["per_query_inputs"]):
        ref_q = g[f"call{i}_query_points"]
        assert q.shape == ref_q.shape
        close(q, ref_q, rtol=1e-5, atol=1e-5)
        assert np.array_equal(q[0, :, 0].long().numpy(), ref_q[0, :, 0].astype(np.int64))
    ref = g["traj_e"]
    assert np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max() < 1e-4
    close(r["vis_e_as_prob"], g["vis_e_as_prob"], rtol=0, atol=1e-3)


def test_updateformer_hidden_384(golden):
    """The class-default hidden_size=384 (mvtracker.py:101); the shipped config uses 256."""
    g = golden("updateformer_h384")
    cfg = O.TrackerConfig(hidden_size=384)
    W384 = O.make_weights(cfg, seed=0)
    close(O.update_former(W384, T(g["x"]), cfg), g["out"], rtol=1e-4, atol=1e-6)


def window_corr_c128_inputs(g):
    rng = np.random.default_rng(int(g["fmaps_seed"]))
    fm = rng.standard_normal(tuple(int(x) for x in g["fmaps_shape"])).astype(np.float32)
    assert abs(float(np.abs(fm).sum()) - float(g["fmaps_checksum"])) < 1e-3 * float(g["fmaps_checksum"])
    return fm


def test_window_corr_c128(golden):
    g = golden("window_corr_c128")
    fm = window_corr_c128_inputs(g)
    pyr = O.window_corr_pyramid(T(fm), 4)
    close(O.window_corr_sample(pyr, T(g["targets"]), T(g["coords"]), 4), g["out_r4"], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------- metrics post-processing (f4)
METRIC_KW = {"3d": dict(distance_thresholds=[0.05, 0.1, 0.2, 0.4, 0.8], survival_distance_threshold=0.5, static_threshold=0.01,
                        dynamic_threshold=0.1, very_dynamic_threshold=2.0),
             "2d": dict(distance_thresholds=[1, 2, 4, 8, 16], survival_distance_threshold=50, static_threshold=None,
                        dynamic_threshold=None, very_dynamic_threshold=None)}


def check_metrics_against_golden(g, name, results, per_track, tol=0.011):
    """results: column -> {metric: value}; the reference table is rounded to 2 decimals (DataFrame.round(2))."""
    cols, idx, table = list(g[f"{name}_columns"]), list(g[f"{name}_index"]), g[f"{name}_table"]
    assert sorted(results) == sorted(cols), (sorted(results), sorted(cols))
    for ci, col in enumerate(cols):
        assert sorted(results[col]) == sorted(idx)
        for ri, metric in enumerate(idx):
            ref, got = table[ri, ci], results[col][metric]
            assert (np.isnan(ref) and np.isnan(got)) or abs(ref - got) <= tol, (col, metric, ref, got)
    n_checked = 0
    for key in g.files:
        if key.startswith(f"{name}_pt__"):
            _, col, k = key.split("__")
            ref, got = g[key], np.asarray(per_track[col][k])
            assert ref.shape == got.shape, key
            if k == "indices":
                assert np.array_equal(ref, got)  # integer track ids: bit-exact
            else:
                ok = (np.isnan(ref) & np.isnan(got)) | (np.abs(ref - got) <= tol + 1e-4 * np.abs(ref))
                assert ok.all(), (key, ref[~ok][:4], got[~ok][:4])
            n_checked += 1
    assert n_checked > 10


@pytest.mark.parametrize("name", ["3d", "2d"])
def test_metrics_oracle(golden, name):
    from oracle import metrics_oracle as MO
    g = golden("metrics_eval")
    res, pt = MO.evaluate_predictions(g[f"{name}_gt"], g[f"{name}_vis"], g[f"{name}_pred"], g[f"{name}_pocc"], g[f"{name}_qp"],
                                      **METRIC_KW[name])
    check_metrics_against_golden(g, name, res, pt)


EVAL3D_CASES = ["kubric", "dexycb", "panoptic_noquery", "tapvid2d", "ablation2d"]


def check_evaluate_3dpt_against_golden(g, name, got, tol=0.011):
    """The flat ``{prefix}/model__{metric}__{point_type}`` dict against the REFERENCE's (tests/golden/make_golden_evaluate3dpt.py,
    evaluator_3dpt.py:62-173): the same key set, every value within ``tol`` (percent-scaled metrics; the reference computes in
    float64 numpy, the product in float32 on the device), NaN where the reference has NaN (empty point-type masks)."""
    keys = [str(k) for k in g[f"{name}_keys"]]
    assert sorted(got) == keys, (sorted(set(keys) ^ set(got))[:5])
    for k, ref in zip(keys, g[f"{name}_values"]):
        v = float(got[k])
        assert (np.isnan(ref) and np.isnan(v)) or abs(ref - v) <= tol * max(1.0, abs(ref) / 100.0), (name, k, ref, v)


@pytest.mark.parametrize("name", EVAL3D_CASES)
def test_evaluate_3dpt_oracle_vs_reference(golden, name):
    from oracle import metrics_oracle as MO
    g = golden("evaluate_3dpt")
    qp = g[f"{name}_qp"] if bool(g[f"{name}_with_query"][0]) else None
    got = MO.evaluate_3dpt(g[f"{name}_gt"], g[f"{name}_vis"], g[f"{name}_pred"], g[f"{name}_pvis"], str(g[f"{name}_setting"][0]),
                           float(g[f"{name}_upscale"][0]), qp)
    check_evaluate_3dpt_against_golden(g, name, got)


def test_predictor_uniform_support_points(golden, W):
    """num_uniformly_sampled_pts > 0 (evaluation_predictor_3dpt.py:147-190, 417-429) against the reference: the sampler makes the
    same two draws in the same order (same CPU generator state -> the recorded points, bit for bit), and the oracle fed those
    points reproduces the model's query rows and the tracks."""
    from mvtracker_amd.predictor import get_uniformly_sampled_pts
    g = golden("predictor_uniform_pts")
    torch.manual_seed(int(g["torch_seed"]))
    sp = get_uniformly_sampled_pts(6, 12, (96, 160), device="cpu")[0]
    assert np.array_equal(sp.numpy(), g["sampled_pts"])
    clip = synth.make_clip(int(g["clip_seed"]), V=2, T=12, H=96, W=160, N=4)
    a = [T(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    r = O.predictor_forward(W, CFG, *a, interp_shape=None, grid_size=2, n_iters=2, uniform_pts=T(g["sampled_pts"]))
    q = torch.cat([a[2], r["support_points"]], 1)
    close(q, g["model_query_points"], rtol=1e-5, atol=1e-5)
    ref = g["traj_e"]
    assert np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max() < 1e-4
    close(r["vis_e_as_prob"], g["vis_e_as_prob"], rtol=0, atol=1e-3)


CORR_OPT_CASES = {"g4_xyz": dict(corr_n_groups=4, corr_add_neighbor_offset=True, corr_add_neighbor_xyz=True),
                  "g2_nooffset": dict(corr_n_groups=2, corr_add_neighbor_offset=False, corr_add_neighbor_xyz=False),
                  "g1_k8_xyz": dict(corr_n_groups=1, corr_neighbors=8, corr_add_neighbor_offset=True, corr_add_neighbor_xyz=True)}


def corr_opts_clip(g):
    return synth.make_clip(int(g["clip_seed"]), V=2, T=18, H=128, W=128, N=10, late_queries=True, query_frames=(3, 7))


@pytest.mark.parametrize("name", list(CORR_OPT_CASES))
def test_corr_options_end_to_end_oracle(golden, name):
    """The reference's non-default correlation layouts (grouped dots, no offsets, neighbour coordinates; mvtracker.py:130-149,
    832-846) end to end: the oracle against the reference fixture (tests/golden/make_golden_corr_opts.py), two windows, late queries."""
    g = golden("e2e_corr_opts")
    cfg = O.TrackerConfig(**CORR_OPT_CASES[name])
    assert cfg.token_dim == int(g[name + "_token_dim"])
    clip = corr_opts_clip(g)
    a = [T(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    with torch.no_grad():
        r = O.tracker_forward(O.make_weights(cfg, 0), cfg, *a, iters=3, knn_mode="exact")
    ref = g[name + "_traj"]
    assert np.abs(r["traj_e"].numpy() - ref).max() / np.abs(ref).max() < 1e-4
    close(r["vis_e"], g[name + "_vis"], rtol=0, atol=1e-3)
