"""single_point evaluation mode (evaluation_predictor_3dpt.py:191-277: one forward per query, each with its local + global support
grids) on a synthetic clip: ms per predictor call.   python tools/time_single_point.py [n_queries]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import synth  # noqa: E402
from mvtracker_amd.predictor import EvaluationPredictor  # noqa: E402
from mvtracker_amd.tracker import MVTracker  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.to("cuda")
m.precision = "bf16"
clip = synth.make_clip(5, V=4, T=24, H=384, W=512, N=nq)
a = {k: torch.from_numpy(v).cuda() for k, v in clip.items()}
pred = EvaluationPredictor(m, interp_shape=None, single_point=True, n_iters=4)
for _ in range(2):
    r = pred(rgbs=a["rgbs"], depths=a["depths"], query_points_3d=a["query_points"], intrs=a["intrs"], extrs=a["extrs"])
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    r = pred(rgbs=a["rgbs"], depths=a["depths"], query_points_3d=a["query_points"], intrs=a["intrs"], extrs=a["extrs"])
torch.cuda.synchronize()
print(f"single_point, {nq} queries, 4 views x 24 frames x 384x512, bf16: {(time.perf_counter() - t0) / K * 1e3:.1f} ms per call")
