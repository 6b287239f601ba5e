"""Random clip shapes through the whole forward, every window / iteration teacher-forced against the oracle on the device's own
state (tests/test_gpu_e2e.py::_check_forward_trace: kNN indices bit-exact, sampled correlation rows), plus finiteness and
run-to-run bit-identity.  A robustness sweep for the GPU box, not part of the test suite:

    python tests/checks/fuzz_forward.py [n_configs] [first_seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402
import test_gpu_e2e as E  # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to(E.DEV)
fails = 0
for k in range(n_cfg):
    rng = np.random.default_rng(1000 + seed0 + k)
    V = int(rng.integers(1, 6))
    wide = os.environ.get("FUZZ_WIDE", "0") != "0"  # also clips shorter than two windows, sizes in multiples of 8, 1..6 iterations
    T = int(rng.integers(1, 40)) if wide else int(rng.integers(13, 32))          # (default: at least two windows)
    H = int(rng.integers(8, 48)) * 8 if wide else int(rng.integers(4, 24)) * 16  # (default 64 .. 368 in multiples of 16: odd pyramid levels included)
    W = int(rng.integers(8, 60)) * 8 if wide else int(rng.integers(4, 30)) * 16
    iters = int(rng.integers(1, 7)) if wide else 4
    big = os.environ.get("FUZZ_BIG", "0") != "0"  # benchmark-sized clips: up to 6 views of 720 x 1 280, up to 4 096 queries
    if big:
        V = int(rng.integers(2, 7))
        H, W = int(rng.integers(16, 46)) * 16, int(rng.integers(16, 81)) * 16
    N = int(rng.choice([1, 2, 7, 33, 100, 341, 512, 700, 1500]))
    if os.environ.get("FUZZ_BIG", "0") != "0":
        N = int(rng.choice([512, 1024, 2048, 4096]))
    prec = str(rng.choice(["fp32", "bf16", "bf16x3"] if os.environ.get("FUZZ_BF16X3", "0") != "0" else ["fp32", "bf16"]))
    late = bool(rng.integers(2))
    inval = float(rng.choice([0.0, 0.03]))
    tag = f"cfg {k}: V={V} T={T} {H}x{W} N={N} {prec} late={late} invalid={inval} iters={iters}"
    t0 = time.time()
    try:
        clip = synth.make_clip(2000 + seed0 + k, V=V, T=T, H=H, W=W, N=N, late_queries=late, invalid_frac=inval, frame_period=4 if big else None)
        a = E.args_of(clip, E.DEV)
        with E._with_precision(m, prec):
            r1 = m(*a, iters=iters)
            t1, v1 = r1["traj_e"].clone(), r1["vis_e"].clone()
            m.check_finite()
            assert bool(torch.isfinite(t1).all()) and bool(torch.isfinite(v1).all())
            r2 = m(*a, iters=iters)
            assert torch.equal(t1, r2["traj_e"]) and torch.equal(v1, r2["vis_e"]), "two runs differ"
            if len(m.last_windows) == 0:
                # the reference's loop `while ind < T - S // 2` (mvtracker.py:537) runs no window when the first query frame is within S/2 = 6
                # frames of the clip's end (any clip of <= 6 frames): its outputs stay zero, and so do the oracle's and ours
                assert float(t1.abs().max()) == 0.0 and float(v1.abs().max()) == 0.0
                print(f"ok   {tag}: no window (first query frame >= T - S/2): zero outputs, as in the reference", flush=True)
                continue
            if len(m.last_windows) >= 2 and m.last_windows[0][1] > 0:
                w = E._check_forward_trace(m, a, n_sample=max(2, min(32, N)), iters=iters)
            else:  # one window (or nothing carried): the same teacher-forced rows, without the carried-track sample
                store = m.build_frame_store(a[0][0], a[1][0], a[3][0], a[4][0])
                tr = []
                m(*a, iters=iters, frame_store=store, trace=tr)
                torch.cuda.synchronize()
                w = 0.0
                for (w0, p1), wt in zip(m.last_windows, tr):
                    sample = torch.randperm(p1, generator=torch.Generator().manual_seed(7))[:max(1, min(32, p1))]
                    for it in range(iters):
                        w = max(w, E._check_iteration_rows(m, store, w0, wt, it, sample))
        print(f"ok   {tag}: fcorr rows max abs err {w:.2e} ({time.time() - t0:.1f} s)", flush=True)
    except ValueError as e:
        if "fewer than corr_neighbors" not in str(e):
            raise
        print(f"ok   {tag}: refused (frames too small for the pyramid)", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
