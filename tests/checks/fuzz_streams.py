"""The multi-stream paths on random clips, each expected BIT-IDENTICAL to its serialised form: (a) MVTracker.forward with the
second-stream encoder / pre-searches against the same call with every hand-over synchronised (sync_debug) and against
overlap_encoder = False; (b) EvaluationPredictor(single_point=True) on 1 stream against 2..8 streams.  A race shows up as a difference.

    python tests/checks/fuzz_streams.py [n_configs] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.predictor import EvaluationPredictor  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

DEV = "cuda:0"
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to(DEV)
fails = 0
for k in range(n_cfg):
    rng = np.random.default_rng(6000 + seed + k)
    V, T = int(rng.integers(1, 5)), int(rng.integers(13, 40))
    H, W = int(rng.integers(8, 30)) * 16, int(rng.integers(8, 34)) * 16
    N = int(rng.choice([3, 33, 200, 700]))
    prec = str(rng.choice(["fp32", "bf16"]))
    tag = f"cfg {k}: V={V} T={T} {H}x{W} N={N} {prec}"
    try:
        clip = synth.make_clip(7000 + seed + k, V=V, T=T, H=H, W=W, N=N, late_queries=bool(rng.integers(2)))
        a = [torch.from_numpy(clip[kk]).to(DEV) for kk in ("rgbs", "depths", "query_points", "intrs", "extrs")]
        m.precision = prec
        outs = []
        for mode in ("default", "sync_debug", "no_overlap", "default"):
            m.sync_debug = mode == "sync_debug"
            m.overlap_encoder = mode != "no_overlap"
            r = m(*a, iters=3)
            outs.append((r["traj_e"].clone(), r["vis_e"].clone()))
        m.sync_debug, m.overlap_encoder = False, True
        torch.cuda.synchronize()
        for i, mode in enumerate(("sync_debug", "no_overlap", "default again"), 1):
            assert torch.equal(outs[0][0], outs[i][0]) and torch.equal(outs[0][1], outs[i][1]), f"forward: default vs {mode} differ"
        # single_point: a few queries, local + global support grids, 1 stream against several
        nq = min(N, int(rng.integers(2, 7)))
        q = a[2][:, :nq].contiguous()
        pred = EvaluationPredictor(m, interp_shape=None, grid_size=int(rng.integers(1, 4)), local_grid_size=int(rng.integers(2, 5)), local_extent=20,
                                   single_point=True, n_iters=2)
        res = []
        for ns in (1, int(rng.integers(2, 9)), 1):
            pred.single_point_streams = ns
            r = pred(rgbs=a[0], depths=a[1], query_points_3d=q, intrs=a[3], extrs=a[4])
            res.append((r["traj_e"].clone(), r["vis_e_as_prob"].clone(), ns))
        torch.cuda.synchronize()
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), f"single_point: 1 vs {res[1][2]} streams differ"
        assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1]), "single_point: two 1-stream runs differ"
        print(f"ok   {tag}: forward x4 identical; single_point {nq} queries, 1 vs {res[1][2]} streams identical", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
