"""Golden vector for the view assignment of MonocularToMultiViewAdapter (reference
mvtracker/models/core/monocular_baselines.py:630-680).  Run in the build container only (imports /root/reference):

    python tests/golden/make_golden_adapter.py

The wrapped 2-D tracker is out of scope; a recording stand-in receives, per view, the queries the adapter assigned to
that view, from which the integer view index of every query is reconstructed."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.import_reference()
sys.modules["mvtracker.utils.visualizer_mp4"].Visualizer = type("Visualizer", (), {})  # imported by name, never used here
import mvtracker.models.core.monocular_baselines as mb  # noqa: E402
from mvtracker_amd import synth  # noqa: E402

clip = synth.make_clip(71, V=4, T=8, H=96, W=128, N=200, late_queries=True, query_frames=(2, 5))
q = clip["query_points"].copy()
rng = np.random.default_rng(3)
q[0, :40, 1:] += rng.normal(0, 0.6, size=(40, 3)).astype(np.float32)   # some queries off every surface / outside some views
q[0, 40:50, 1:] *= -3.0                                                 # behind most cameras
seen = {}


class Recorder(torch.nn.Module):
    def forward(self, rgbs, depths, intrs, extrs, queries, queries_with_z, queries_xyz_worldspace):
        v = len(seen)
        seen[id(rgbs)] = queries_xyz_worldspace.clone()
        Tn, n = rgbs.shape[0], queries.shape[0]
        return {"traj_2d": torch.zeros(Tn, n, 2), "vis": torch.ones(Tn, n), "traj_3d_worldspace": torch.zeros(Tn, n, 3)}


order = []
rec = Recorder()
orig = rec.forward


def fwd(**kw):
    order.append(kw["queries_xyz_worldspace"].clone())
    return orig(**kw)


rec.forward = fwd
ad = mb.MonocularToMultiViewAdapter(rec)
tt = lambda a: torch.from_numpy(np.asarray(a))
with torch.no_grad():
    try:
        ad(tt(clip["rgbs"]), tt(clip["depths"]), tt(q), tt(clip["intrs"]), tt(clip["extrs"]), save_debug_logs=False)
    except Exception as e:  # the code after the per-view calls is not needed for the assignment
        print("adapter stopped after the per-view calls:", type(e).__name__, str(e)[:80])
# reconstruct the per-query view index: views are visited in order, views without queries are skipped
N = q.shape[1]
view = np.full(N, -1, np.int64)
qa = q[0]
v_iter = iter(order)
remaining = set(range(N))
calls = list(order)
# every call lists its queries in ascending query order (boolean mask); match rows exactly
ci = 0
for v in range(clip["rgbs"].shape[1]):
    if ci >= len(calls):
        break
    rows = calls[ci].numpy()
    idx = [i for i in sorted(remaining) if any((rows[:, 1:] == qa[i, 1:]).all(1) & (rows[:, 0] == qa[i, 0].astype(np.int64)))]
    # a view's call must account for exactly its rows; otherwise this view had no queries and the call belongs to a later view
    if len(idx) >= len(rows) and len(rows) > 0:
        take = []
        for r in rows:
            for i in idx:
                if i not in take and (qa[i, 1:] == r[1:]).all() and int(qa[i, 0]) == int(r[0]):
                    take.append(i)
                    break
        if len(take) == len(rows):
            view[take] = v
            remaining -= set(take)
            ci += 1
assert ci == len(calls) and (view >= 0).all(), (ci, len(calls), int((view < 0).sum()))
np.savez(os.path.join(HERE, "adapter_view_assignment.npz"), seed=71, V=4, T=8, H=96, W=128, N=200, query_points=q, best_view=view)
print("views:", np.bincount(view, minlength=4))
