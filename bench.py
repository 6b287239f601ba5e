#!/usr/bin/env python3
"""Headline benchmark: query-points x frames / second of the multi-view tracking forward path.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full tracker call (encode 4 views x 24 frames of 512x512, build the frame store,
3 sliding windows x 4 refinement iterations, 1024 queries) on synthetic inputs already resident in HBM
(BASELINE.json config "4-view 24-frame 512x512 @1024 queries"; one corr_gather_dot launch covers the 4 pyramid
levels of one refinement iteration = 4 x 12288 units x 9164 B = 450 MB of algorithmic traffic).  With N > 1 every rank tracks its own
1024-query shard (weak scaling; shard = independent forward) and the frames are
encoded once across the node: rank r encodes frames r, r+N, ... and the level-0 feature maps are
all-gathered over RCCL/xGMI before the refinement loop, which contains no collective.

Prints ONE JSON line (rank 0).  `roofline` is measured live for the HBM-bound gather-dot correlation
kernel with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the
reference's algorithm, its CPU fallback kNN) on a bounded sample of the same workload on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def corr_algorithmic_bytes(rows, K=16, C=128):
    """per (frame, track, level) K rows x C x 4 B gathered + K x 12 B neighbour xyz
    + C x 4 B target + 12 B coord read, K x 16 B written = 9164 B at K=16, C=128 (fp32)."""
    return rows * (K * C * 4 + K * 12 + C * 4 + 12 + K * 16)


def cpu_baseline(args):
    """Oracle (port of the reference algorithm, CPU fallback kNN = cdist+topk) on a bounded sample."""
    from mvtracker_amd import synth
    from oracle import mvt_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, args.cpu_threads))  # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(cores)
    V, T, H, W, N = 4, 12, args.cpu_hw, args.cpu_hw, 256  # the workload's views / image size, one window, a quarter of the queries
    clip = synth.make_clip(1234, V=V, T=T, H=H, W=W, N=N)
    cfg = O.TrackerConfig()
    Wt = O.make_weights(cfg, 0)
    a = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    t0 = time.time()
    with torch.no_grad():
        O.tracker_forward(Wt, cfg, *a, iters=4, knn_mode="cdist")
    dt = time.time() - t0
    return {"value": N * T / dt, "unit": "query-points*frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.tracker_forward, {V} views x {T} frames x {H}x{W}, {N} queries, 1 window x 4 iters, fp32, "
                      f"{dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--queries", type=int, default=1024, help="queries per GPU")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="bf16",
                    help="matrix-core arithmetic of convs/linears; BASELINE.json quotes this config in bf16")
    ap.add_argument("--cpu-hw", type=int, default=512)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("MVT_FORCE_SHARDED"):  # (the env switch rehearses the sharded code path on one GPU)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from mvtracker_amd import hip, synth
    from mvtracker_amd.parallel import ShardedTracker
    from mvtracker_amd.tracker import MVTracker

    model = MVTracker(hidden_size=256).eval()
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev)
    if args.precision:
        model.precision = args.precision
    V, T, HW, Nq = args.views, args.frames, args.size, args.queries
    clip = synth.make_clip(1234, V=V, T=T, H=HW, W=HW, N=Nq * world)  # same clip on every rank
    a = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}
    runner = ShardedTracker(model)

    # live per-launch timing of the correlation kernel (HIP events on the launch stream)
    events = []
    real_corr = hip.corr_gather_dot
    timing = {"on": False}

    def timed_corr(*cargs, **ckw):
        if not timing["on"]:
            return real_corr(*cargs, **ckw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_corr(*cargs, **ckw)
        e1.record()
        events.append((e0, e1, cargs[7] * cargs[8]))  # rows = N * S

    hip.corr_gather_dot = timed_corr

    def step():
        return runner(a["rgbs"], a["depths"], a["query_points"], a["intrs"], a["extrs"], iters=args.iters, gather_output=False)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timing["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timing["on"] = False
    model.check_finite()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * Nq * T / (dt / args.steps)
        full = [(e0.elapsed_time(e1), rows) for e0, e1, rows in events if rows == Nq * model.S]
        kern_ms = float(np.mean([m for m, _ in full])) if full else float("nan")
        alg = model.corr_n_levels * corr_algorithmic_bytes(Nq * model.S, model.corr_neighbors, model.latent_dim)
        achieved = alg / (kern_ms * 1e-3) / 1e9 if full else float("nan")
        # HBM traffic of the roofline kernel: PMC measurement committed under profiles/ (tools/pmc_traffic.sh; counters cannot
        # be collected from inside the timed run).  Only quoted for the workload it was measured on.
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_corr_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("algorithmic_bytes_per_launch") == alg:
                traffic = tj["traffic_bytes_per_launch"]
        out = {
            "metric": "query-points*frames/sec, 4-view 24-frame 512x512 @1024 queries",
            "value": value, "unit": "query-points*frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (split-precision bf16 MFMA, f32 accumulate, f32-grade)", "bf16": "bf16"}[model.precision],
            "data": "synthetic",
            "config": {"workload": f"{V}-view {T}-frame {HW}x{HW}, {Nq} queries per GPU, corr K=16 x 4 levels, iters={args.iters}, "
                                   f"3 windows, random-init weights (seeded recipe)",
                       "queries_total": Nq * world, "parallelism": f"query-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "corr_gather_dot_kernel<32>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": kern_ms, "launches_timed": len(full),
                         # launches of the last window run alone; earlier ones share HBM with the encoder of the later
                         # frames on the second stream (the average above includes that contention)
                         "min_launch_ms": float(np.min([m for m, _ in full])) if full else None},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1 or os.environ.get("MVT_FORCE_SHARDED"):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
