"""PCIe-inclusive rate of the C3 workload: host (pinned) clip -> device -> tracker call, per step.
uint8 frames (the sample files' storage type, 75 MB) + fp32 depth (101 MB) vs fp32 frames (302 MB) + depth."""
import sys, time, torch
sys.path.insert(0, ".")
from mvtracker_amd import synth
from mvtracker_amd.tracker import MVTracker
import numpy as np
dev = torch.device("cuda:0")
model = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev); model.precision = "bf16"
clip = synth.make_clip(1234, V=4, T=24, H=512, W=512, N=1024)
for name, rgb in (("uint8 frames", torch.from_numpy(clip["rgbs"].astype(np.uint8))), ("fp32 frames", torch.from_numpy(clip["rgbs"]))):
    host = [rgb.pin_memory()] + [torch.from_numpy(clip[k]).pin_memory() for k in ("depths", "query_points", "intrs", "extrs")]
    mb = sum(t.numel() * t.element_size() for t in host) / 1e6
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = [t.to(dev, non_blocking=True) for t in host]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model(*a, iters=4)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{name}: {mb:.0f} MB host->device in {1e3 * (t1 - t0):.2f} ms ({mb / 1e3 / (t1 - t0):.1f} GB/s), tracker {1e3 * (t2 - t1):.2f} ms, "
          f"PCIe-inclusive {1024 * 24 / (t2 - t0):.0f} query-points*frames/s")
