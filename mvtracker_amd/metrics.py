"""Metrics post-processing of the reference on the device.

``evaluate_predictions`` / ``evaluate_3dpt`` mirror mvtracker/evaluation/metrics.py:303-406 and
evaluation/evaluator_3dpt.py:62-173 (same arguments, same result keys, values rounded to two decimals like the reference's
``DataFrame.round(2)``).  The per-track work -- distances, TAP-Vid occlusion / position / Jaccard counts, median / average /
final trajectory error, survival, point movement -- runs in ONE kernel launch (``mvt_track_metrics``, one wave per track);
what is left is a handful of masked means over the (N, 11 + 2K) result, done with device tensor ops and read back once.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip

# evaluation/evaluator_3dpt.py:89-128: (distance thresholds, survival threshold, static, dynamic, very dynamic, point dim)
SETTINGS = {
    "kubric-multiview": ([0.05, 0.1, 0.2, 0.4, 0.8], 0.5, 0.01, 0.1, 2.0, 3),
    "dexycb-multiview": ([0.01, 0.02, 0.05, 0.1, 0.2], 0.1, 0.01, 0.1, 0.5, 3),
    "panoptic-multiview": ([0.05, 0.10, 0.20, 0.40], 1.0, None, None, None, 3),
    "tapvid2d": ([1, 2, 4, 8, 16], 50, None, None, None, 2),
    "2dpt_ablation": ([1, 2, 4, 8, 16], 50, 1, 1, 50, 2),
}
_FIXED = ["occlusion_accuracy", "occlusion_accuracy_for_vis0", "occlusion_accuracy_for_vis1", "average_jaccard",
          "average_pts_within_thresh", "mte_visible", "ate_visible", "fde_visible", "survival"]


def _dev(a, dtype, dev):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=dev, dtype=dtype).contiguous()


def _names(thresholds) -> list:
    th = [f"{float(np.float32(t)):.2f}" for t in thresholds]  # the reference formats the float32 threshold tensor (metrics.py:148)
    return _FIXED + [f"pts_within_{t}" for t in th] + [f"jaccard_{t}" for t in th]


@torch.no_grad()
def per_track_metrics(gt_tracks, gt_visibilities, pred_tracks, pred_occluded, query_points, distance_thresholds, survival_distance_threshold,
                      device="cuda"):
    """(names, table (N, len(names)) fp32 fractions, movement (N,), visible frames (N,), query frames (N,)) on ``device``."""
    dev = torch.device(device) if not isinstance(gt_tracks, torch.Tensor) or not gt_tracks.is_cuda else gt_tracks.device
    gt = _dev(gt_tracks, torch.float32, dev)
    T, N, D = gt.shape
    with hip.device_guard(gt):
        pr = _dev(pred_tracks, torch.float32, dev)
        vis = _dev(gt_visibilities, torch.uint8, dev)
        occ = _dev(pred_occluded, torch.uint8, dev)
        qt = _dev(query_points, torch.float32, dev)[:, 0].long().int().contiguous()  # float32 -> .long(), as the reference's tensors
        K = len(distance_thresholds)
        out = torch.empty(N, 11 + 2 * K, device=dev)
        hip.track_metrics(gt, pr, vis, occ, qt, T, N, D, distance_thresholds, survival_distance_threshold, out)
    return _names(distance_thresholds), out[:, 2:], out[:, 0], out[:, 1], qt


@torch.no_grad()
def evaluate_predictions(gt_tracks, gt_visibilities, pred_tracks, pred_occluded, query_points=None,
                         distance_thresholds=(0.01, 0.02, 0.04, 0.08, 0.16), survival_distance_threshold=0.5, static_threshold=0.01,
                         dynamic_threshold=0.1, very_dynamic_threshold=2.0, device="cuda") -> Tuple[Dict, Dict]:
    """metrics.py:303-406.  Returns (results: column -> {metric: percent rounded to 2 decimals}, per_track: column -> {metric +
    "_per_track": (n,) percent values, "indices": track ids}) -- the contents of the reference's two DataFrames."""
    is_t = isinstance(gt_tracks, torch.Tensor)
    T, N, D = gt_tracks.shape
    if query_points is None:  # metrics.py:316-320: the first visible frame is the query
        gv = gt_visibilities if is_t else torch.from_numpy(np.asarray(gt_visibilities))
        gtt = gt_tracks if is_t else torch.from_numpy(np.asarray(gt_tracks))
        qt0 = gv.to(torch.uint8).argmax(0)
        query_points = torch.cat([qt0[:, None].to(gtt.dtype), gtt[qt0, torch.arange(N, device=gtt.device)]], -1)
    names, table, movement, nvis, qt = per_track_metrics(gt_tracks, gt_visibilities, pred_tracks, pred_occluded, query_points,
                                                         list(distance_thresholds), survival_distance_threshold, device)
    types = [("any", torch.ones_like(movement, dtype=torch.bool))]
    if static_threshold is not None:
        types.append(("static", movement < static_threshold))
    if dynamic_threshold is not None:
        types.append(("dynamic", movement > dynamic_threshold))
    if very_dynamic_threshold is not None:
        types.append(("very_dynamic", movement > very_dynamic_threshold))
    mask_a = nvis >= 2  # at least two visible frames, the first one is the query (metrics.py:346)
    cols, means, counts, vsum = [], [], [], []
    for name, mask_b in types:
        m = mask_a & mask_b
        cols.append((name, m))
        sel = torch.where(m[:, None], table, torch.full_like(table, float("nan")))
        means.append(torch.nanmean(sel, dim=0))
        counts.append(m.sum())
        vsum.append((nvis * m).sum())
    # ONE readback for everything the tables need
    packed = torch.cat([torch.stack(means).reshape(-1), torch.stack(counts).float(), torch.stack(vsum).float()]).cpu().numpy()
    nm = len(names)
    results: Dict[str, Dict[str, float]] = {}
    per_track: Dict[str, Dict[str, np.ndarray]] = {}
    table_h = None
    for i, (name, m) in enumerate(cols):
        cnt = int(round(float(packed[len(cols) * nm + i])))
        if cnt == 0:
            continue
        col = f"all_{name}"
        results[col] = {k: float(packed[i * nm + j]) * 100 for j, k in enumerate(names)}
        results[col]["n"] = cnt / N * 100
        results[col]["v"] = float(packed[len(cols) * nm + len(cols) + i]) / cnt / T * 100
        if table_h is None:
            table_h = table.cpu().numpy()
        mh = m.cpu().numpy()
        per_track[col] = {k + "_per_track": np.round(table_h[mh, j] * 100, 2) for j, k in enumerate(names)}
        per_track[col]["indices"] = np.where(mh)[0]
    if "all_static" in results and "all_dynamic" in results:  # metrics.py:394-397
        results["all_dynamic-static-mean"] = {k: (results["all_dynamic"][k] + results["all_static"][k]) / 2 for k in results["all_static"]}
    results = {c: {k: float(np.round(v, 2)) for k, v in d.items()} for c, d in results.items()}
    return results, per_track


@torch.no_grad()
def evaluate_3dpt(gt_tracks, gt_visibilities, pred_tracks, pred_visibilities, evaluation_setting, track_upscaling_factor, query_points=None,
                  prefix="3dpt", verbose=False, add_per_track_results=True, device="cuda") -> Dict:
    """evaluation/evaluator_3dpt.py:62-173: the flat ``{prefix}/model__{metric}__{point_type}`` dict the evaluator logs."""
    T, N, D = gt_tracks.shape
    assert tuple(gt_tracks.shape) == tuple(pred_tracks.shape)
    assert tuple(gt_visibilities.shape) == (T, N) and tuple(pred_visibilities.shape) == (T, N)
    if evaluation_setting not in SETTINGS:
        raise NotImplementedError(evaluation_setting)
    th, surv, st, dy, vd, dim = SETTINGS[evaluation_setting]
    assert D == dim
    as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    gt, pr, gv, pv = as_t(gt_tracks), as_t(pred_tracks), as_t(gt_visibilities), as_t(pred_visibilities)
    if query_points is None:  # :80-83
        qt0 = gv.to(torch.uint8).argmax(0)
        qp = torch.cat([qt0[:, None].to(gt.dtype), gt[qt0, torch.arange(N, device=gt.device)]], -1)
    else:
        qp = as_t(query_points)
    qp = torch.cat([qp[:, 0:1], qp[:, 1:] * track_upscaling_factor], -1)
    res, per_track = evaluate_predictions(gt * track_upscaling_factor, gv.bool(), pr * track_upscaling_factor, ~pv.bool(), qp, th, surv, st, dy,
                                          vd, device=device)
    out: Dict = {}
    for point_type in ["dynamic-static-mean", "dynamic", "very_dynamic", "static", "any"]:  # :147-151
        col = f"all_{point_type}"
        if col not in res:
            continue
        for metric in sorted(res[col]):
            out[f"{prefix}/model__{metric}__{point_type}"] = res[col][metric]
    if add_per_track_results:
        out[f"{prefix}/model__per_track_results"] = per_track
    return out
