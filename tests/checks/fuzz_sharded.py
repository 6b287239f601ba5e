"""The multi-GPU code path (per-rank frame block -> RCCL all-gather of the level-0 features, second exchange on the side stream ->
frame store handed to the forward -> gathered outputs) rehearsed on ONE GPU with a one-rank RCCL group (MVT_FORCE_SHARDED=1) on random
clips: bit-identical to the direct call.

    python tests/checks/fuzz_sharded.py [n_configs] [seed]
"""
import os
import sys

os.environ["MVT_FORCE_SHARDED"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.parallel import ShardedTracker  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

DEV = "cuda:0"
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to(DEV)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device(DEV))
fails = 0
try:
    runner = ShardedTracker(m)
    for k in range(n_cfg):
        rng = np.random.default_rng(8000 + seed + k)
        V, T = int(rng.integers(1, 5)), int(rng.integers(13, 40))
        H, W = int(rng.integers(8, 30)) * 16, int(rng.integers(8, 34)) * 16
        N = int(rng.choice([1, 3, 33, 200, 700]))
        prec = str(rng.choice(["fp32", "bf16"]))
        tag = f"cfg {k}: V={V} T={T} {H}x{W} N={N} {prec}"
        try:
            clip = synth.make_clip(9000 + seed + k, V=V, T=T, H=H, W=W, N=N, late_queries=bool(rng.integers(2)))
            a = [torch.from_numpy(clip[kk]).to(DEV) for kk in ("rgbs", "depths", "query_points", "intrs", "extrs")]
            m.precision = prec
            ref = m(*a, iters=3)
            rt, rv = ref["traj_e"].clone(), ref["vis_e"].clone()
            for rep in range(2):
                out = runner(*a, iters=3)
                torch.cuda.synchronize()
                assert torch.equal(out["traj_e"], rt) and torch.equal(out["vis_e"], rv), f"sharded path differs from the direct call (run {rep})"
            runner.check_finite_collective()
            print(f"ok   {tag}", flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
finally:
    dist.destroy_process_group()
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
