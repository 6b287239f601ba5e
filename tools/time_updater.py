"""Time one updater call (mvt_updateformer_forward through MVTracker.update_former) alone on the chip, bf16 mode.
    python tools/time_updater.py [n_tracks] [reps]      (tuning switches through the environment)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to("cuda")
m.precision = "bf16"
x = torch.randn(1, n, 12, 581, generator=torch.Generator().manual_seed(n)).cuda()
out = m.update_former(x)
for _ in range(3):
    m.update_former(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    m.update_former(x)
e1.record()
torch.cuda.synchronize()
print(f"updater n={n} fuse_attention={m.fuse_attention}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call; checksum {out.double().abs().sum().item():.6e}")
