"""Constructor options of the reference model (mvtracker.py:94-149) drawn at random -- sliding_window_len 8 / 12 / 16, corr_neighbors 8 / 16,
grouped dots, with / without neighbour offsets and coordinates, hidden_size 256 / 384 -- on random clips, fp32 and bf16: finite,
run-to-run bit-identical, every window / iteration teacher-forced against the oracle (kNN indices bit-exact, sampled correlation
rows in the option's layout), and one updater call of the forward against the oracle's update_former with the same configuration
(fp32: 2e-5 of the output scale; bf16: the 1.1 x rule of _bf16_stage_check).

    python tests/checks/fuzz_model_options.py [n_configs] [seed]
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
from oracle import mvt_oracle as O  # noqa: E402

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
for k in range(n_cfg):
    rng = np.random.default_rng(11000 + seed0 + k)
    S = int(rng.choice([8, 12, 16]))
    opts = dict(sliding_window_len=S, corr_neighbors=int(rng.choice([8, 16])), corr_n_groups=int(rng.choice([1, 2, 4])),
                corr_add_neighbor_offset=bool(rng.integers(2)), corr_add_neighbor_xyz=bool(rng.integers(2)))
    hidden = int(rng.choice([256, 256, 384]))
    V, T = int(rng.integers(1, 5)), int(rng.integers(S + 1, 3 * S + 4))
    H, W = int(rng.integers(6, 20)) * 16, int(rng.integers(6, 24)) * 16
    N = int(rng.choice([5, 33, 200, 450, 700]))
    prec = str(rng.choice(["fp32", "bf16"]))
    tag = f"cfg {k}: {opts} hidden={hidden} | V={V} T={T} {H}x{W} N={N} {prec}"
    t0 = time.time()
    try:
        cfg = O.TrackerConfig(hidden_size=hidden, **opts)
        m = MVTracker(hidden_size=hidden, **opts).eval()
        assert m.updateformer_input_dim == cfg.token_dim
        sd = synth.make_state_dict({kk: tuple(v.shape) for kk, v in m.state_dict().items()}, seed=0)
        m.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()}, strict=True)
        m = m.to(E.DEV)
        m.precision = prec
        Wc = O.make_weights(cfg, seed=0)
        clip = synth.make_clip(12000 + seed0 + k, V=V, T=T, H=H, W=W, N=N, late_queries=bool(rng.integers(2)))
        a = E.args_of(clip, E.DEV)
        r1 = m(*a, iters=3)
        t1, v1 = r1["traj_e"].clone(), r1["vis_e"].clone()
        m.check_finite()
        assert bool(torch.isfinite(t1).all()) and bool(torch.isfinite(v1).all())
        r2 = m(*a, iters=3)
        assert torch.equal(t1, r2["traj_e"]) and torch.equal(v1, r2["vis_e"]), "two runs differ"
        if len(m.last_windows) == 0:
            print(f"ok   {tag}: no window", flush=True)
            continue
        store = m.build_frame_store(a[0][0], a[1][0], a[3][0], a[4][0])
        tr = []
        m(*a, iters=3, frame_store=store, trace=tr)
        torch.cuda.synchronize()
        w = 0.0
        for (w0, p1), wt in zip(m.last_windows, tr):
            sample = torch.randperm(p1, generator=torch.Generator().manual_seed(7))[:max(1, min(24, p1))]
            for it in range(3):
                w = max(w, E._check_iteration_rows(m, store, w0, wt, it, sample))
        # one updater call of the last window against the oracle with this configuration
        tok = tr[-1]["tokens"][1].cpu().float()[None]
        got = tr[-1]["delta"][1].cpu().float().numpy()[None]
        with torch.no_grad():
            ref = O.update_former(Wc, tok, cfg).numpy()
            if prec == "bf16":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    ac = O.update_former(Wc, tok, cfg).float().numpy()
                E._bf16_stage_check(f"      updater n = {tok.shape[1]}", got, ref, ac, (3e-2, 2.5e-2))
            else:
                err = np.abs(got - ref).max() / np.abs(ref).max()
                assert err < 2e-5, f"fp32 updater against the oracle: {err:.2e}"
        print(f"ok   {tag}: fcorr rows {w:.2e} ({time.time() - t0:.1f} s)", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
