"""Probe: steady-state step time when the NEXT clip's frame store (geometry + whole encoder) is built on a second stream while the
current clip's windows run (two different synthetic clips alternate; nothing is reused).  Prints ms per step for the plain
sequence and for the pipelined one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

dev = torch.device("cuda:0")
model = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev)
model.precision = "bf16"
clips = []
for seed in (1234, 4321):
    c = synth.make_clip(seed, V=4, T=24, H=512, W=512, N=1024)
    clips.append({k: torch.from_numpy(v).to(dev) for k, v in c.items()})
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = sys.argv[2] if len(sys.argv) > 2 else "whole"   # whole: the whole encoder ahead; first: only the first window's frames


def run_plain(i):
    a = clips[i % 2]
    return model(a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"], iters=4)


pref = torch.cuda.Stream(device=dev)


def prefetch(i):
    a = clips[i % 2]
    f32 = lambda t: t.to(torch.float32).contiguous()
    main = torch.cuda.current_stream(dev)
    with torch.cuda.stream(pref):
        st = model.build_frame_store(f32(a["rgbs"][0]), f32(a["depths"][0]), f32(a["intrs"][0]), f32(a["extrs"][0]), t0=0)
        ev = torch.cuda.Event()
        ev.record(pref)
    for v in st.values():
        for t in (v if isinstance(v, (list, tuple)) else [v]):
            if torch.is_tensor(t):
                t.record_stream(main)
    st["pending"] = [(0, ev)]
    return st


def run_piped(i, store):
    nxt = prefetch(i + 1)
    a = clips[i % 2]
    r = model(a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"], iters=4, frame_store=store)
    return r, nxt


for i in range(3):
    run_plain(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    run_plain(i)
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / K * 1e3
ref = [run_plain(i)["traj_e"].clone() for i in range(2)]
store = prefetch(0)
for i in range(3):
    _, store = run_piped(i, store)
torch.cuda.synchronize()
store = prefetch(K % 2 if False else 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    r, store = run_piped(i, store)
torch.cuda.synchronize()
piped = (time.perf_counter() - t0) / K * 1e3
same = torch.equal(r["traj_e"], ref[(K - 1) % 2])
print(f"plain {plain:.2f} ms/step   pipelined {piped:.2f} ms/step   last result identical to the plain forward: {same}")
