"""mvtracker_amd.metrics (mvt_track_metrics: one wave per track) on random clip lengths, track counts, visibility patterns and
evaluation settings against the CPU restatement of the reference's evaluation/metrics.py (oracle/metrics_oracle.py): the per-track
table for every track and evaluate_3dpt's flat dictionary.

    python tests/checks/fuzz_metrics.py [n_configs] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvtracker_amd import metrics  # noqa: E402
from oracle import metrics_oracle as MO  # noqa: E402

DEV = "cuda:0"
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
SETTINGS = ["kubric-multiview", "dexycb-multiview", "panoptic-multiview", "tapvid2d", "2dpt_ablation"]
fails = 0
for k in range(n_cfg):
    rng = np.random.default_rng(4000 + seed + k)
    T = int(rng.choice([1, 2, 3, 8, 24, 63, 64, 65, 100, 129, 200]))
    N = int(rng.choice([1, 2, 5, 33, 64, 65, 300]))
    D = 3
    pvis_rate = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
    noise = float(rng.choice([0.0, 0.01, 0.05, 0.5]))
    setting = SETTINGS[int(rng.integers(len(SETTINGS)))]
    if setting in ("tapvid2d", "2dpt_ablation"):
        D = 2
    tag = f"cfg {k}: T={T} N={N} D={D} visible {pvis_rate} noise {noise} {setting}"
    try:
        gt = (rng.uniform(-1, 1, (1, N, D)) + np.cumsum(rng.standard_normal((T, N, D)) * 0.02, 0)).astype(np.float32)
        vis = rng.uniform(size=(T, N)) < pvis_rate
        qt = rng.integers(0, max(1, T // 2 + 1), size=N)
        vis[qt, np.arange(N)] = True
        pred = (gt + rng.standard_normal((T, N, D)) * noise).astype(np.float32)
        pvis = vis ^ (rng.uniform(size=(T, N)) < 0.1)
        qp = np.concatenate([qt[:, None].astype(np.float32), gt[qt, np.arange(N)]], -1).astype(np.float32)
        thr = [0.01, 0.02, 0.05, 0.1, 0.2] if D == 3 else [1, 2, 4, 8, 16]
        surv = 0.5 if D == 3 else 50
        gt_vis = vis & (np.arange(T)[:, None] >= qt[None, :])
        tm = MO.track_metrics(gt, gt_vis, pred, ~pvis, qp, thr, surv)
        names, table, movement, nvis, _ = metrics.per_track_metrics(gt, vis, pred, ~pvis, qp, thr, surv, device=DEV)
        th = table.cpu().numpy()
        for j, nm in enumerate(names):
            ref = tm[nm]
            ok = (np.isnan(ref) & np.isnan(th[:, j])) | (np.abs(ref - th[:, j]) <= 1e-5 * (1 + np.abs(ref)))
            assert ok.all(), (nm, ref[~ok][:3], th[~ok, j][:3])
        assert np.allclose(movement.cpu().numpy(), MO.point_movement(gt, gt_vis), rtol=1e-5, atol=1e-6), "movement"
        assert np.array_equal(nvis.cpu().numpy().astype(np.int64), gt_vis.sum(0)), "visible counts"
        if True:
            got = metrics.evaluate_3dpt(gt, vis, pred, pvis, setting, 2.0, qp, add_per_track_results=False, device=DEV)
            ref = MO.evaluate_3dpt(gt, vis, pred, pvis, setting, 2.0, qp)
            assert sorted(got) == sorted(ref), "keys"
            for kk in ref:
                assert (np.isnan(ref[kk]) and np.isnan(got[kk])) or abs(ref[kk] - got[kk]) <= 0.011, (kk, ref[kk], got[kk])
        print(f"ok   {tag}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
