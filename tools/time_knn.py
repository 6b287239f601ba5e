"""Time the seeded kNN search (mvt_knn_search_levels) on a C3-shaped synthetic cloud: all levels in one launch and each level alone.

    python tools/time_knn.py            (MVT_KNN_Q=1|2|4|8 selects the queries per wave)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda"
N, S, K, T, V = 1024, 12, 16, 12, 4
torch.manual_seed(0)
levels = []
for hw in (128, 64, 32, 16):
    P = V * hw * hw
    sc = 128 // hw
    ys, xs = torch.meshgrid(torch.arange(hw).float(), torch.arange(hw).float(), indexing="ij")
    base = torch.stack([xs * 0.03 * sc, ys * 0.03 * sc, 0.3 * torch.sin(xs * 0.1 * sc)], -1).reshape(1, 1, hw * hw, 3)
    xyz = torch.zeros(T, P, 4, device=dev)
    xyz[..., :3] = (base + torch.rand(T, V, hw * hw, 3) * 0.01 * sc + torch.arange(V).view(1, V, 1, 1) * 0.007).reshape(T, P, 3).to(dev)
    box = torch.empty(T, (P + 63) // 64, 8, device=dev)
    hip.tile_aabb(xyz, P, T, box, (hw, hw))
    gbox = torch.empty(T, ((P + 63) // 64 + 63) // 64, 8, device=dev)
    hip.tile_group_aabb(box, P, T, gbox)
    levels.append(dict(xyz=xyz, P=P, box=box, grid=(hw, hw), gbox=gbox if os.environ.get("NO_GBOX") is None else None))
q = (torch.rand(N, S, 3) * torch.tensor([128 * 0.03, 128 * 0.03, 0.3])).to(dev)
for lv in levels:  # exact unseeded neighbours of q = the seeds
    nseg = 1
    keys = torch.empty(N * S * nseg * K, device=dev, dtype=torch.int64)
    hip.knn_scan(lv["xyz"], lv["P"], q, N, S, 0, 1, T, K, nseg, keys, box=lv["box"], grid=lv["grid"])
    lv["seed"] = torch.empty(N, S, K, device=dev, dtype=torch.int32)
    hip.knn_merge(keys, N, S, K, nseg, lv["P"], lv["seed"])
q2 = q + torch.randn_like(q) * 0.004


def run(sel):
    lvs = [dict(xyz=levels[i]["xyz"], P=levels[i]["P"], box=levels[i]["box"], grid=levels[i]["grid"], seed_idx=levels[i]["seed"], gbox=levels[i]["gbox"],
                idx_out=torch.empty(N, S, K, device=dev, dtype=torch.int32)) for i in sel]
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.knn_search_levels(lvs, q2, N, S, 0, 1, T, K, K)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    return min(ts[1:])




def run_unseeded(sel):
    lvs = [dict(xyz=levels[i]["xyz"], P=levels[i]["P"], box=levels[i]["box"], grid=levels[i]["grid"], seed_idx=None, gbox=levels[i]["gbox"],
                idx_out=torch.empty(N, S, K, device=dev, dtype=torch.int32)) for i in sel]
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.knn_search_levels(lvs, q2, N, S, 0, 1, T, K, 0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 200)
    return min(ts[1:])


print(f"unseeded: all levels {run_unseeded([0, 1, 2, 3]):.1f} us; " + "  ".join(f"L{i} {run_unseeded([i]):.1f}" for i in range(4)))
print(f"Q={os.environ.get('MVT_KNN_Q', '2')}: all levels {run([0, 1, 2, 3]):.1f} us; " + "  ".join(f"L{i} {run([i]):.1f}" for i in range(4)))
