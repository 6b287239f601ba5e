"""EvaluationPredictor's input stage (nearest resize of frames / depths, rescaled intrinsics, support grids unprojected through
bilinearly sampled depth; evaluation_predictor_3dpt.py:59-120) on random clip sizes, interp shapes and grid settings: what the
predictor hands to MVTracker.forward against the oracle's predictor_prepare (deterministic: no neighbour ranking involved).

    python tests/checks/fuzz_predictor.py [n_configs] [seed]
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
from oracle import mvt_oracle as O  # noqa: E402

DEV = "cuda:0"
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to(DEV)
fails = 0
for k in range(n_cfg):
    rng = np.random.default_rng(900 + seed + k)
    V, T = int(rng.integers(1, 5)), int(rng.integers(8, 20))
    H, W = int(rng.integers(6, 30)) * 8, int(rng.integers(6, 40)) * 8
    N = int(rng.integers(1, 40))
    interp = None if rng.integers(4) == 0 else (int(rng.integers(8, 30)) * 16, int(rng.integers(8, 36)) * 16)
    grid = int(rng.integers(0, 7))
    ngv = int(rng.integers(1, 4))
    tag = f"cfg {k}: V={V} T={T} {H}x{W} N={N} interp={interp} grid={grid} grids/view={ngv}"
    try:
        clip = synth.make_clip(3000 + seed + k, V=V, T=T, H=H, W=W, N=N, late_queries=bool(rng.integers(2)))
        c = {kk: torch.from_numpy(v) for kk, v in clip.items()}
        ref_rgbs, ref_depths, ref_intrs, ref_support = O.predictor_prepare(c["rgbs"], c["depths"], c["query_points"], c["intrs"], c["extrs"],
                                                                           interp, grid, ngv)
        ref_q = torch.cat([c["query_points"], ref_support], 1)
        calls = []
        orig = m.forward

        def spy(rgbs, **kw):
            calls.append({kk: (v.detach().float().cpu() if torch.is_tensor(v) else v) for kk, v in dict(kw, rgbs=rgbs).items()})
            return orig(rgbs, **kw)

        m.forward = spy
        try:
            pred = EvaluationPredictor(m, interp_shape=interp, grid_size=grid, n_grids_per_view=ngv, n_iters=1)
            pred(rgbs=c["rgbs"].to(DEV), depths=c["depths"].to(DEV), query_points_3d=c["query_points"].to(DEV), intrs=c["intrs"].to(DEV),
                 extrs=c["extrs"].to(DEV))
        finally:
            m.forward = orig
        assert len(calls) == 1
        kw = calls[0]
        assert torch.equal(kw["rgbs"], ref_rgbs.float()), "resized frames differ"
        assert torch.equal(kw["depths"], ref_depths.float()), "resized depths differ"
        ei = (kw["intrs"] - ref_intrs).abs().max().item()
        q = kw["query_points"]
        assert q.shape == ref_q.shape, (q.shape, ref_q.shape)
        eq = ((q - ref_q).abs().max() / ref_q.abs().max().clamp_min(1.0)).item()
        assert ei < 1e-4 and eq < 1e-5, (ei, eq)
        print(f"ok   {tag}: {q.shape[1]} queries, intrinsics {ei:.1e}, query points {eq:.1e}", flush=True)
    except ValueError as e:
        if "fewer than corr_neighbors" not in str(e):
            raise
        print(f"ok   {tag}: refused with a clear error (frames too small for the pyramid, as in the reference's kNN)", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
