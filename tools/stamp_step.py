"""In-kernel stamps of one block-kernel instantiation inside a whole C3 tracking step (diagnostic build, see tools/stamp_block.py):

    hipcc ... -DMVT_STAMPS '-DMVT_STAMP_SEL=(NMB == 1 && MODE == 2 && ATT == 0)' -DMVT_STAMP_WG1=1 -c block_fused.hip ; link
    MVT_LIB=<that .so> python tools/stamp_step.py

The buffer keeps the LAST launch of the selected instantiation."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

lib = ctypes.CDLL(os.environ["MVT_LIB"])
dev = torch.device("cuda:0")
model = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev)
model.precision = "bf16"
clip = synth.make_clip(1234, V=4, T=24, H=512, W=512, N=1024)
a = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}
inputs = (a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"])
model(*inputs, iters=4)
torch.cuda.synchronize()
assert lib.mvt_debug_clear_stamps() == 0
model(*inputs, iters=4)
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 64, dtype=np.uint64)
assert lib.mvt_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
buf = buf.reshape(2, 8, 64).astype(np.int64)
names = {0: "start", 1: "pre-attn", 2: "tile staged", 3: "barrier", 4: "out-proj", 5: "x added", 6: "LN1", 30: "mlp end", 31: "store x", 63: "end"}
names.update({56: "ctx summed", 57: "ctx LN", 58: "ctx proj", 59: "ctx next"})
for c in range(4):
    names.update({8 + 5 * c: f"c{c} begin", 9 + 5 * c: f"c{c} fc1", 10 + 5 * c: f"c{c} gelu+st", 11 + 5 * c: f"c{c} barrier", 12 + 5 * c: f"c{c} fc2"})
for q in range(3):
    names.update({32 + 8 * q: f"p{q} begin", 33 + 8 * q: f"p{q} LN", 39 + 8 * q: f"p{q} end"})
    for b in range(5):
        names[34 + 8 * q + b] = f"p{q} blk{b}"
for wg in range(2):
    if not (buf[wg] > 0).any():
        continue
    t0 = buf[wg][buf[wg] > 0].min()
    print(f"workgroup slot {wg}: cycles since the first stamp / delta, per wave")
    prev = {w: None for w in range(8)}
    for i in [i for i in range(64) if (buf[wg, :, i] > 0).any()]:
        cells = []
        for w in range(8):
            v = buf[wg, w, i]
            if v <= 0:
                cells.append("      -      ")
                continue
            d = v - prev[w] if prev[w] is not None else 0
            prev[w] = v
            cells.append(f"{v - t0:6d}/{d:5d}")
        print(f"{i:2d} {names.get(i, ''):12s} " + " ".join(cells))
