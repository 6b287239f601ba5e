"""Find seeds whose product-vs-oracle run has no kNN neighbour flip (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvtracker_amd import synth
from mvtracker_amd.tracker import MVTracker
from oracle import mvt_oracle as O
dev = "cuda:0"
cfg = O.TrackerConfig(); W = O.make_weights(cfg, 0)
m = MVTracker(hidden_size=256).eval(); m.load_state_dict(W, strict=True); m.to(dev)
torch.set_num_threads(16)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    clip = synth.make_clip(seed, V=3, T=20, H=128, W=160, N=24, late_queries=True)
    cpu = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    tr = []
    r = m(*[t.to(dev) for t in cpu], iters=4, trace=tr)
    otr = {}
    ro = O.tracker_forward(W, cfg, *cpu, iters=4, knn_mode="exact", trace=otr)
    flips = 0
    for wt, ow in zip(tr, otr["windows"]):
        for it in range(len(wt["knn_idx"])):
            for lvl in range(4):
                flips += int((wt["knn_idx"][it][lvl].cpu().long() != ow["knn_idx"][it * 4 + lvl].permute(1, 0, 2)).sum())
    ref = ro["traj_e"]
    print(seed, "flips", flips, "tracks rel", f"{((r['traj_e'].cpu()-ref).abs().max()/ref.abs().max()).item():.2e}",
          "vis", f"{(m.last_vis_logits.cpu()-ro['vis_logits']).abs().max().item():.2e}", flush=True)
