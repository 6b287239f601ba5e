"""CPU restatement (numpy) of the reference's metrics post-processing -- SURVEY.md section 8f rank 4.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED for ``evaluate_predictions`` and everything below it
(tests/golden/make_golden_metrics.py imports mvtracker/evaluation/metrics.py from /root/reference -- it needs numpy, pandas and
torch only -- and tests/test_oracle_golden.py re-checks this file against the recorded outputs).  ``evaluate_3dpt`` (the thin
wrapper in evaluation/evaluator_3dpt.py:62-173: per-dataset threshold table + flattening of the table into a dict) cannot be
imported here (its module pulls in rerun, imageio, tensorboard, ...) and is restated from the source text: parity UNPINNED for
that wrapper only.

Per-track metrics are computed for EVERY track (NaN where the reference's subset would not contain the track); the reference
computes them per point-type subset, which gives the same per-track numbers because no metric couples tracks.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

# evaluation/evaluator_3dpt.py:89-128: distance thresholds, survival threshold, static / dynamic / very-dynamic movement thresholds
SETTINGS = {
    "kubric-multiview": ([0.05, 0.1, 0.2, 0.4, 0.8], 0.5, 0.01, 0.1, 2.0, 3),
    "dexycb-multiview": ([0.01, 0.02, 0.05, 0.1, 0.2], 0.1, 0.01, 0.1, 0.5, 3),
    "panoptic-multiview": ([0.05, 0.10, 0.20, 0.40], 1.0, None, None, None, 3),
    "tapvid2d": ([1, 2, 4, 8, 16], 50, None, None, None, 2),
    "2dpt_ablation": ([1, 2, 4, 8, 16], 50, 1, 1, 50, 2),
}


def thresh_name(t: float) -> str:
    """metrics.py:148,157: keys are formatted from the float32 threshold tensor with two decimals."""
    return f"{float(np.float32(t)):.2f}"


def track_metrics(gt_tracks, gt_vis, pred_tracks, pred_occ, query_points, distance_thresholds, survival_distance_threshold):
    """compute_metrics + compute_tapvid_metrics(query_mode="first") (metrics.py:10-58, 61-171) for every track.

    gt_tracks / pred_tracks (T,N,D), gt_vis (T,N) bool ALREADY masked to frames >= the query frame, pred_occ (T,N) bool,
    query_points (N,1+D).  Returns name -> (N,) float32 arrays (fractions, not percent)."""
    T, N, _ = gt_tracks.shape
    gt = gt_tracks.astype(np.float32)
    pr = pred_tracks.astype(np.float32)
    qt = query_points[:, 0].astype(np.float32).astype(np.int64)  # torch .long() of the float32 query tensor
    gt_occ = ~gt_vis
    fr = np.arange(T)[:, None]
    evalp = np.ones((T, N), bool)
    evalp[np.clip(qt, 0, T - 1), np.arange(N)] = False  # :119-122 (the query frame is not evaluated)
    evalp &= ~(fr < qt[None, :])                        # :125-128
    f32 = np.float32
    out: Dict[str, np.ndarray] = {}
    with np.errstate(divide="ignore", invalid="ignore"):
        agree = pred_occ == gt_occ
        out["occlusion_accuracy"] = (agree & evalp).sum(0).astype(f32) / evalp.sum(0).astype(f32)
        out["occlusion_accuracy_for_vis0"] = (agree & gt_occ & evalp).sum(0).astype(f32) / (gt_occ & evalp).sum(0).astype(f32)
        out["occlusion_accuracy_for_vis1"] = (agree & ~gt_occ & evalp).sum(0).astype(f32) / (~gt_occ & evalp).sum(0).astype(f32)
        diff = pr - gt
        dist = np.sqrt((diff * diff).sum(-1, dtype=np.float32)).astype(f32)
        jac, pts = [], []
        for th in distance_thresholds:
            within = dist < f32(th)
            vis_pts = (~gt_occ & evalp).sum(0).astype(f32)
            pw = (within & ~gt_occ & evalp).sum(0).astype(f32) / vis_pts
            tp = (within & ~pred_occ & ~gt_occ & evalp).sum(0).astype(f32)
            fp = (((~within & ~pred_occ) | (~pred_occ & gt_occ)) & evalp).sum(0).astype(f32)
            jc = tp / (vis_pts + fp)
            out[f"pts_within_{thresh_name(th)}"] = pw
            out[f"jaccard_{thresh_name(th)}"] = jc
            pts.append(pw)
            jac.append(jc)
        out["average_jaccard"] = np.stack(jac, -1).mean(-1, dtype=f32)
        out["average_pts_within_thresh"] = np.stack(pts, -1).mean(-1, dtype=f32)
        # metrics.py:27-45: trajectory errors over visible frames at / after the query frame (the query frame itself included)
        d = dist.copy()
        d[~gt_vis] = np.nan
        d[fr < qt[None, :]] = np.nan
        mte = np.full(N, np.nan, f32)
        for n in range(N):
            v = np.sort(d[~np.isnan(d[:, n]), n])
            if len(v):
                mte[n] = v[(len(v) - 1) // 2]  # torch.nanmedian: the LOWER of the two middle values
        out["mte_visible"] = mte
        cnt = (~np.isnan(d)).sum(0)
        out["ate_visible"] = (np.nansum(d.astype(np.float64), 0) / cnt).astype(f32)
        last = np.argmax(gt_vis * np.arange(T)[:, None], 0)
        out["fde_visible"] = d[last, np.arange(N)]
        failed = (d > f32(survival_distance_threshold)) & gt_vis
        fidx = np.where(failed.any(0), failed.argmax(0), T)
        out["survival"] = ((fidx - qt) / (T - qt)).astype(f32)
    return out


def point_movement(gt_tracks, gt_vis):
    """metrics.py:327-330: path length over the visible frames of a track."""
    T, N, _ = gt_tracks.shape
    mv = np.zeros(N)
    for n in range(N):
        tr = gt_tracks[gt_vis[:, n], n, :]
        mv[n] = np.linalg.norm(tr[1:] - tr[:-1], axis=-1).sum()
    return mv


def evaluate_predictions(gt_tracks, gt_visibilities, pred_tracks, pred_occluded, query_points=None,
                         distance_thresholds=(0.01, 0.02, 0.04, 0.08, 0.16), survival_distance_threshold=0.5, static_threshold=0.01,
                         dynamic_threshold=0.1, very_dynamic_threshold=2.0):
    """metrics.py:303-406.  Returns (results: column -> {metric: value rounded to 2 decimals}, per_track: column -> {metric:
    (n_sel,) percent values rounded to 2 decimals, "indices": track ids})."""
    T, N, _ = gt_tracks.shape
    if query_points is None:  # :316-320
        qt = np.argmax(gt_visibilities, axis=0)
        query_points = np.concatenate([qt[:, None], gt_tracks[qt, np.arange(N)]], axis=-1)
    later = np.arange(T)[:, None] >= query_points[:, 0][None, :]
    gt_vis = gt_visibilities.copy() * later
    movement = point_movement(gt_tracks, gt_vis)
    types = [("any", np.ones(N, bool))]
    if static_threshold is not None:
        types.append(("static", movement < static_threshold))
    if dynamic_threshold is not None:
        types.append(("dynamic", movement > dynamic_threshold))
    if very_dynamic_threshold is not None:
        types.append(("very_dynamic", movement > very_dynamic_threshold))
    mask_a = gt_vis.sum(0) >= 2  # :346
    tm = track_metrics(gt_tracks, gt_vis, pred_tracks, pred_occluded, query_points, distance_thresholds, survival_distance_threshold)
    results, per_track = {}, {}
    for name, mask_b in types:
        m = mask_a & mask_b
        col = f"all_{name}"
        if m.sum() == 0:
            continue
        results[col] = {}
        per_track[col] = {}
        for k, v in tm.items():
            vv = v[m]
            with np.errstate(invalid="ignore"):
                results[col][k] = float(np.nanmean(vv.astype(np.float32), dtype=np.float32)) * 100 if (~np.isnan(vv)).any() else float("nan")
            per_track[col][k + "_per_track"] = np.round(vv.astype(np.float32) * 100, 2)
        results[col]["n"] = m.sum() / N * 100
        results[col]["v"] = gt_vis[:, m].sum() / m.sum() / T * 100
        per_track[col]["indices"] = np.where(m)[0]
    if "all_static" in results and "all_dynamic" in results:  # :394-397
        results["all_dynamic-static-mean"] = {k: (results["all_dynamic"][k] + results["all_static"][k]) / 2 for k in results["all_static"]}
    results = {c: {k: float(np.round(v, 2)) for k, v in d.items()} for c, d in results.items()}
    return results, per_track


def evaluate_3dpt(gt_tracks, gt_visibilities, pred_tracks, pred_visibilities, evaluation_setting, track_upscaling_factor,
                  query_points=None, prefix="3dpt"):
    """evaluation/evaluator_3dpt.py:62-173 without the logging and the per-track table."""
    T, N, D = gt_tracks.shape
    if query_points is None:  # :80-83
        qt = gt_visibilities.argmax(axis=0)
        query_points = np.concatenate([qt[:, None], gt_tracks[qt, np.arange(N), :]], axis=-1)
    th, surv, st, dy, vd, dim = SETTINGS[evaluation_setting]
    assert D == dim
    qp = np.concatenate([query_points[:, 0:1], query_points[:, 1:] * track_upscaling_factor], axis=-1)
    res, _ = evaluate_predictions(gt_tracks * track_upscaling_factor, gt_visibilities, pred_tracks * track_upscaling_factor,
                                  ~pred_visibilities, qp, th, surv, st, dy, vd)
    out = {}
    for point_type in ["dynamic-static-mean", "dynamic", "very_dynamic", "static", "any"]:  # :147-151
        col = f"all_{point_type}"
        if col not in res:
            continue
        for metric in sorted(res[col]):
            out[f"{prefix}/model__{metric}__{point_type}"] = res[col][metric]
    return out
