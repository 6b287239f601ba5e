"""cProfile of the host side of one C3 tracking call (bf16): where the Python / launch time goes."""
import cProfile
import os
import pstats
import sys

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
clip = synth.make_clip(1234, V=4, T=24, H=512, W=512, N=1024)
a = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}
inputs = (a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"])
for _ in range(3):
    model(*inputs, iters=4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    model(*inputs, iters=4)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
st.sort_stats("tottime").print_stats(25)
