import sys, time, torch
sys.path.insert(0, "/root/repo")
import numpy as np
from mvtracker_amd import synth
from mvtracker_amd.tracker import MVTracker
dev = torch.device("cuda:0")
model = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev); model.precision = "bf16"
clip = synth.make_clip(1234, V=4, T=24, H=512, W=512, N=1024)
a = [torch.from_numpy(clip[k]).to(dev) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
for _ in range(2):
    model(*a, iters=4)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    model(*a, iters=4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
