"""The bf16 updater (mvt_updateformer_forward: every form of the block kernels -- split path, 32-row small-M forms, 64-row tiles, deferred
pass 2, key-split attention) at track counts on and around its form thresholds and at random ones, against the oracle's
EfficientUpdateFormer on the same tokens: the rule of tests/test_gpu_e2e.py::_bf16_stage_check (at most 10 % worse than the oracle
under bf16 autocast, in max and in mean, plus an absolute cap).

    python tests/checks/fuzz_updater.py [n_random] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402
import test_gpu_e2e as E  # noqa: E402
from oracle import mvt_oracle as O  # noqa: E402

n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to(E.DEV)
m.precision = "bf16"
W = O.make_weights(E.CFG, seed=0)
rng = np.random.default_rng(seed)
# thresholds: split path below 4 096 point rows (342 tracks), time block on 32-row tiles up to 448 tracks, point<-virtual block on
# 32-token tiles for 342..672 tracks, 64-row tiles above; 5-track (60-row) tiles: counts that leave partial tiles
counts = [1, 2, 5, 31, 64, 341, 342, 343, 447, 448, 449, 512, 671, 672, 673, 680, 1023, 1025] + [int(x) for x in rng.integers(3, 1600, n_rand)]
D = 581
fails = 0
for n in counts:
    g = torch.Generator().manual_seed(1000 + n)
    x = torch.randn(1, n, 12, D, generator=g) * 0.5
    try:
        out = m.update_former(x.to(E.DEV))
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = O.update_former(W, x, E.CFG).numpy()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ac = O.update_former(W, x, E.CFG).float().numpy()
        E._bf16_stage_check(f"n = {n:5d}", out.cpu().numpy(), ref, ac, (3e-2, 2.5e-2))
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL n = {n}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{len(counts) - fails} / {len(counts)} track counts passed")
sys.exit(1 if fails else 0)
