#!/usr/bin/env python3
"""Offline demo of the MI355X tracker: the reference's demo.py flow (load a sample NPZ -> track -> save the result NPZ) without
its remote pieces (torch.hub / HuggingFace downloads, Rerun logging, depth estimators).

    python demo_amd.py --sample-path data_sample.npz --checkpoint mvtracker_200000_june2025.pth --save-npz tracks.npz
    python demo_amd.py --synthetic --precision bf16            # seeded synthetic clip, seeded random weights (no files needed)

Mirrors reference demo.py: sample layout :650, 922-929; temporal / spatial subsampling :905-944 (``--temporal_stride``,
``--spatial_downsample``); ``--random_query_points`` :967-993 (512 queries drawn from the depth of frame 0 inside a cylinder);
the predictor call :1004-1010 (bf16 = the demo's autocast arithmetic); the result file :1086-1121.  The wall time of the predictor
call is reported with the evaluator's convention (frames / second, evaluation/evaluator_3dpt.py:496-523)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def random_queries(depths, intrs, extrs, num_queries=512, t0=0, xy_radius=12.0, z_min=-1.0, z_max=10.0, seed=0):
    """demo.py:967-993: unproject every pixel of frame t0 (all views), keep the points inside the cylinder, draw num_queries."""
    from mvtracker_amd import hip
    _, V, T, _, H, W = depths.shape
    dev = depths.device
    kinv = torch.empty(V * T, 9, device=dev)
    einv = torch.empty(V * T, 12, device=dev)
    hip.invert_cameras(intrs[0].reshape(V * T, 9).contiguous(), extrs[0].reshape(V * T, 12).contiguous(), kinv, einv, V * T)
    ds = depths[0, :, :, 0].permute(1, 0, 2, 3).contiguous()  # (T,V,H,W): the frame store's depth layout at stride 1
    xyz = torch.empty(T, V, H, W, 4, device=dev)
    hip.unproject(ds, kinv, einv, xyz, V, T, H, W, 1, 0)
    pts = xyz[t0].reshape(-1, 4)[:, :3]
    r2 = pts[:, 0] ** 2 + pts[:, 1] ** 2
    pool = pts[(r2 <= xy_radius ** 2) & (pts[:, 2] >= z_min) & (pts[:, 2] <= z_max) & (ds[t0].reshape(-1) > 0)]
    assert pool.shape[0] > 0, "cylinder mask removed all points; increase the radius or the z range"
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randperm(pool.shape[0], generator=g)[:num_queries].to(dev)
    q = pool[idx]
    return torch.cat([torch.full((q.shape[0], 1), float(t0), device=dev), q], 1)[None]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--sample-path", help="sample NPZ (rgbs, depths, intrs, extrs[, query_points])")
    src.add_argument("--synthetic", action="store_true", help="seeded synthetic clip (4 views x 24 frames x 384x512, 256 queries)")
    ap.add_argument("--checkpoint", help="reference checkpoint (.pth, read with weights_only=True); default: seeded random weights")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="bf16", help="bf16 = the demo's autocast arithmetic")
    ap.add_argument("--temporal_stride", type=int, default=1)
    ap.add_argument("--spatial_downsample", type=int, default=1)
    ap.add_argument("--random_query_points", action="store_true")
    ap.add_argument("--single_point", action="store_true")
    ap.add_argument("--grid-size", type=int, default=5)
    ap.add_argument("--n-iters", type=int, default=4)
    ap.add_argument("--interp-shape", type=int, nargs=2, default=None, metavar=("H", "W"), help="resize like the evaluator (e.g. 384 512)")
    ap.add_argument("--save-npz", help="result file (tracks_3d, visibilities, query_points, camera data)")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("demo_amd.py needs an MI355X: the tracker has no CPU path")
    from mvtracker_amd import sample_io, synth
    from mvtracker_amd.factory import load_mvtracker

    dev = torch.device(args.device)
    torch.cuda.set_device(dev)
    predictor = load_mvtracker(checkpoint=args.checkpoint, device=dev, interp_shape=tuple(args.interp_shape) if args.interp_shape else None,
                               grid_size=args.grid_size, n_iters=args.n_iters, single_point=args.single_point)
    model = predictor.model
    if args.checkpoint is None:
        print("no --checkpoint: seeded random weights (results are meaningless as tracks, the pipeline is the real one)")
        sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        model.to(dev)
    model.precision = args.precision
    if args.synthetic:
        clip = synth.make_clip(3, V=4, T=24, H=384, W=512, N=256, late_queries=True, rgb_dtype=np.uint8)
        s = {k: torch.from_numpy(v)[:, :, ::args.temporal_stride].to(dev) if k != "query_points" else torch.from_numpy(v).to(dev)
             for k, v in clip.items()}
        s["query_points_3d"] = s.pop("query_points")
        if args.temporal_stride > 1:
            s["query_points_3d"][..., 0] = torch.floor(s["query_points_3d"][..., 0] / args.temporal_stride)
    else:
        s = sample_io.load_sample(args.sample_path, device=dev, temporal_stride=args.temporal_stride, spatial_downsample=args.spatial_downsample)
    if args.random_query_points or s["query_points_3d"].shape[1] == 0:
        s["query_points_3d"] = random_queries(s["depths"].float(), s["intrs"], s["extrs"])
    V, T = s["rgbs"].shape[1:3]
    print(f"clip: {V} views x {T} frames x {tuple(s['rgbs'].shape[-2:])}, {s['query_points_3d'].shape[1]} queries, precision {args.precision}")
    call = lambda: predictor(rgbs=s["rgbs"], depths=s["depths"], intrs=s["intrs"], extrs=s["extrs"], query_points_3d=s["query_points_3d"])
    call()  # warm-up (weight packing, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = call()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = s["query_points_3d"].shape[1]
    print(f"predictor call: {1e3 * dt:.1f} ms = {T / dt:.1f} frames/s = {n * T / dt:.0f} query-points*frames/s; "
          f"{int(out['vis_e'].sum())} of {out['vis_e'].numel()} track points visible; NaN guard {'TRIPPED' if predictor.last_nan else 'clean'}")
    if args.save_npz:
        sample_io.save_result(args.save_npz, out["traj_e"], out["vis_e"], s, temporal_stride=args.temporal_stride,
                              spatial_downsample=args.spatial_downsample)
        print("saved", args.save_npz)


if __name__ == "__main__":
    main()
