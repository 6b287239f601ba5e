"""Probe: two independent clips in flight on two HIP streams (forwards enqueued alternately) against one clip after the other.
Prints ms per clip for both."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

dev = torch.device("cuda:0")


def make():
    m = MVTracker(hidden_size=256).eval()
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.to(dev)
    m.precision = "bf16"
    return m


# two model objects (same weights): every piece of per-call scratch and every helper stream is private to its forward
models = [make(), make()]
clips = []
for seed in (1234, 4321):
    c = synth.make_clip(seed, V=4, T=24, H=512, W=512, N=1024)
    clips.append({k: torch.from_numpy(v).to(dev) for k, v in c.items()})
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def run(i, m):
    a = clips[i % 2]
    return m(a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"], iters=4)


for i in range(3):
    run(i, models[0])
    run(i, models[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    run(i, models[0])
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / K * 1e3
ref = run(0, models[0])["traj_e"].clone()
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
import threading


def worker(j, n, out):
    with torch.cuda.stream(streams[j]):
        for i in range(n):
            out[j] = run(2 * i + j, models[j])


for _ in range(2):  # warm-up, then the timed run
    outs = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(j, K // 2, outs)) for j in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    par = (time.perf_counter() - t0) / (2 * (K // 2)) * 1e3
same = torch.equal(outs[0]["traj_e"], ref)
print(f"one clip at a time {seq:.2f} ms/clip   two clips in flight {par:.2f} ms/clip   result identical: {same}")
