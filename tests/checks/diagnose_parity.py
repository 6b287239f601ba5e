"""Stage-by-stage parity report (GPU box): product vs oracle on one seeded clip.

    python tests/checks/diagnose_parity.py --seed 52 --views 3 --frames 20 --height 128 --width 160 --queries 24 --late
Prints, per window / iteration, the number of kNN index mismatches per level and the max errors of the
correlation features, tokens, deltas; then final track / visibility-logit errors."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402
from oracle import mvt_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=52)
ap.add_argument("--views", type=int, default=3)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--height", type=int, default=128)
ap.add_argument("--width", type=int, default=160)
ap.add_argument("--queries", type=int, default=24)
ap.add_argument("--late", action="store_true")
a = ap.parse_args()

dev = "cuda:0"
cfg = O.TrackerConfig()
W = O.make_weights(cfg, 0)
m = MVTracker(hidden_size=256).eval()
m.load_state_dict({k: v for k, v in W.items()}, strict=True)
m.to(dev)
clip = synth.make_clip(a.seed, V=a.views, T=a.frames, H=a.height, W=a.width, N=a.queries, late_queries=a.late)
cpu = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
tr = []
r = m(*[t.to(dev) for t in cpu], iters=4, trace=tr)
otr = {}
ro = O.tracker_forward(W, cfg, *cpu, iters=4, knn_mode="exact", trace=otr)
L = cfg.corr_n_levels
for wi, (wt, ow) in enumerate(zip(tr, otr["windows"])):
    for it in range(len(wt["knn_idx"])):
        mism = []
        for lvl in range(L):
            oi = ow["knn_idx"][it * L + lvl].permute(1, 0, 2)
            mism.append(int((wt["knn_idx"][it][lvl].cpu().long() != oi).sum()))
        fc = (wt["fcorrs"][it].cpu() - ow["fcorrs"][it][0].permute(1, 0, 2)).abs().max().item()
        tk = (wt["tokens"][it].cpu() - ow["tokens"][it][0]).abs().max().item()
        de = (wt["delta"][it].cpu() - ow["delta"][it]).abs().max().item() / ow["delta"][it].abs().max().item()
        print(f"window {wi} iter {it}: kNN mismatches per level {mism}  fcorr {fc:.2e}  tokens {tk:.2e}  delta rel {de:.2e}")
ref = ro["traj_e"]
print("tracks rel err", ((r["traj_e"].cpu() - ref).abs().max() / ref.abs().max()).item())
dv = (m.last_vis_logits.cpu() - ro["vis_logits"]).abs()
print("vis logit max err", dv.max().item(), "count > 1e-3:", int((dv > 1e-3).sum()), "of", dv.numel())
