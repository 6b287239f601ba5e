"""Host-side timeline of one C3 tracking call (bf16): when, relative to the start of forward(), the host sync returns, the first
encoder kernel is enqueued, the first updater call is enqueued, and the call returns -- against the GPU time of the step.

    python tools/host_bubble.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip, synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

dev = torch.device("cuda:0")
model = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev)
model.precision = "bf16"
clip = synth.make_clip(1234, V=4, T=24, H=512, W=512, N=1024)
a = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}
inputs = (a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"])
marks = {}


def tap(mod, name, key):
    real = getattr(mod, name)

    def f(*args, **kw):
        marks.setdefault(key, time.perf_counter())
        return real(*args, **kw)
    setattr(mod, name, f)


tap(hip, "invert_cameras", "first geometry launch")
tap(hip, "rgb_images_to_nhwc4", "first encoder launch")
tap(hip, "corr_gather_dot", "first correlation launch")
tap(hip, "updateformer_forward_tokens", "first updater launch")
for _ in range(3):
    model(*inputs, iters=4)
torch.cuda.synchronize()
for rep in range(3):
    marks.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    model(*inputs, iters=4)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: GPU {e0.elapsed_time(e1):.2f} ms | host: " + ", ".join(f"{k} +{(v - t0) * 1e3:.2f} ms" for k, v in marks.items())
          + f", forward returns +{(t1 - t0) * 1e3:.2f} ms, GPU done +{(t2 - t0) * 1e3:.2f} ms")
