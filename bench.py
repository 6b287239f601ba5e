#!/usr/bin/env python3
"""Headline benchmark: query-points x frames / second of the multi-view tracking forward path.

    python bench.py --gpus 1 --steps 10 --warmup 2 [--config c3|c2|c5shard]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full tracker call on synthetic inputs already resident in HBM: encode the V x T frames, build the frame store,
sliding windows x 4 refinement iterations.  Default workload = BASELINE.json's headline config C3 (4 views x 24 frames x 512x512,
1024 queries, bf16): 3 windows; one corr_gather_dot launch covers the 4 pyramid levels of one refinement iteration = 4 x 12288
units x 4812 B (bf16 feature rows; 9164 B with the fp32 rows of the fp32 / bf16x3 modes) of algorithmic traffic.
With N > 1 every rank tracks its own 1024-query shard (weak scaling; shard = independent forward) and the
frames are encoded once across the node: the V x T images are cut evenly across the ranks (contiguous runs of the frame-major
image list), each rank encodes its run into its slot of the level-0 store and the slots are all-gathered in place over
RCCL/xGMI before the refinement loop, which contains no collective.

Prints ONE JSON line (rank 0).  `roofline` is measured live for the HBM-bound gather-dot correlation kernel with HIP events on
the launch stream; `roofline_mfma` gives the matrix-core fraction of the two MFMA-bound stages (updater transformer, CNN
encoder; flops recomputed from the shapes, HIP events around every call); `cpu_baseline` times the CPU oracle (a port of the
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

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak; fp32 MFMA 157.3
MFMA_F32_TFLOPS = 157.3

CONFIGS = {
    # name: (views, frames, H, W, queries per GPU, precision, extra make_clip kwargs, BASELINE.json config it is)
    "c3": (4, 24, 512, 512, 1024, "bf16", {}, "4-view synthetic 24-frame 512x512, 1024 queries, bf16 (configs[2], the metric's config)"),
    "c2": (3, 24, 384, 512, 512, "fp32", {"invalid_frac": 0.02}, "3-view 24 frames 384x512, 512 query points, fp32 (configs[1])"),
    "c5shard": (6, 64, 720, 1280, 512, "bf16", {"frame_period": 4, "rgb_dtype": np.uint8},
                "one GPU's 512-query shard of the 6-view 64-frame 720p config (configs[4])"),
}


def corr_algorithmic_bytes(rows, K=16, C=128, elem=4):
    """per (frame, track, level) K rows x C x e_f gathered + K x 12 B neighbour xyz + C x e_f target
    + 12 B coord read, K x 16 B written = 9164 B (fp32 rows) / 4812 B (bf16 rows) at K=16, C=128."""
    return rows * (K * C * elem + K * 12 + C * elem + 12 + K * 16)


def updater_flops(n, S=12, h=256, inner=288, mlp=1024, D=581, out=131, nv=64, heads=6, dh=48, depth=6):
    """EfficientUpdateFormer.forward on n tracks (cotracker2/blocks.py:455-494), multiply-adds x 2, from the shapes."""
    Mp, Mv = n * S, nv * S
    M = Mp + Mv
    lin = lambda rows, k, nn: 2.0 * rows * k * nn
    att = lambda groups, nq, nk: 4.0 * groups * heads * nq * nk * dh
    blk = lambda rows: lin(rows, inner, h) + lin(rows, h, mlp) + lin(rows, mlp, h)
    f = lin(Mp, D, h)
    per = (lin(M, h, 3 * inner) + att(n + nv, S, S) + blk(M)                     # time block
           + lin(Mv, h, inner) + lin(Mp, h, 2 * inner) + att(S, nv, n) + blk(Mv)   # virtual <- point
           + lin(Mv, h, 3 * inner) + att(S, nv, nv) + blk(Mv)                      # virtual self
           + lin(Mp, h, inner) + lin(Mv, h, 2 * inner) + att(S, n, nv) + blk(Mp))  # point <- virtual
    f += depth * per
    f += lin(Mp, h, out) + 2 * lin(Mp, out, out)
    return f


def encoder_flops(H, W, C=128):
    """BasicEncoder.forward on one H x W image (spatracker/blocks.py:214-284), multiply-adds x 2."""
    conv = lambda ho, wo, cout, cin, k: 2.0 * ho * wo * cout * cin * k * k
    h2, w2 = H // 2, W // 2
    f = conv(h2, w2, 64, 3, 7)
    f += 4 * conv(h2, w2, 64, 64, 3)
    cin, h, w = 64, h2, w2
    for cout in (96, 128, 128):
        h, w = h // 2, w // 2
        f += conv(h, w, cout, cin, 3) + 3 * conv(h, w, cout, cout, 3) + conv(h, w, cout, cin, 1)
        cin = cout
    f += conv(H // 4, W // 4, 2 * C, 416, 3) + conv(H // 4, W // 4, C, 2 * C, 1)
    return f


def cpu_baseline(args, V, HW):
    """Oracle (port of the reference algorithm, CPU fallback kNN = cdist+topk) on a bounded sample."""
    from mvtracker_amd import synth
    from oracle import mvt_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, args.cpu_threads))  # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(cores)
    H, W = HW
    if args.cpu_hw:
        H = W = args.cpu_hw
    elif H * W > 512 * 512:  # (720p clips: a quarter-resolution sample keeps the CPU leg within ~30 s)
        H, W = H // 2, W // 2
    T, N = 12, 256  # the workload's views / image size, one window, a quarter of the queries
    clip = synth.make_clip(1234, V=V, T=T, H=H, W=W, N=N)
    cfg = O.TrackerConfig()
    Wt = O.make_weights(cfg, 0)
    a = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    t0 = time.time()
    with torch.no_grad():
        O.tracker_forward(Wt, cfg, *a, iters=4, knn_mode="cdist")
    dt = time.time() - t0
    return {"value": N * T / dt, "unit": "query-points*frames/s", "cores": torch.get_num_threads(), "kind": "port",
            # a quarter-size sample (12 of 24 frames, 256 of 1024 queries: ~10 s instead of ~65 s) keeps the default run short;
            # the kNN cost per query-frame is the same, the encoder cost per query-frame is 2x the full clip's
            "sample": f"oracle.tracker_forward, {V} views x {T} frames x {H}x{W}, {N} queries, 1 window x 4 iters, fp32, "
                      f"{dt:.1f} s wall"}


def two_clips_in_flight(model, a, args, MVTracker, synth, dev, V, T, H, W, Nq, clip_kw, prec):
    """Throughput with TWO independent clips in flight on one GPU: two model objects (same weights; every piece of per-call scratch
    and every helper stream is private to its forward), two host threads, two HIP streams, two different clips.  One clip's
    virtual-track chain (<= 96 busy CUs) runs beside the other's 12 k-row blocks.  A serving-layer figure: the reference forward is
    batch 1 (mvtracker.py:275, 503), so this is NEVER the headline `value`."""
    import threading
    m2 = MVTracker(hidden_size=256).eval()
    m2.load_state_dict(model.state_dict(), strict=True)
    m2.to(dev)
    m2.precision = prec
    c2 = synth.make_clip(4321, V=V, T=T, H=H, W=W, N=Nq, **clip_kw)
    b = {k: torch.from_numpy(v).to(dev) for k, v in c2.items()}
    models, clips = [model, m2], [a, b]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    K = max(2, args.steps // 2)

    def call(j):
        c = clips[j]
        return models[j](c["rgbs"], c["depths"], c["query_points"], c["intrs"], c["extrs"], iters=args.iters)

    ref = [call(j)["traj_e"].clone() for j in range(2)]  # one clip at a time (also warms the second model up)
    torch.cuda.synchronize()
    outs = [None, None]

    def worker(j):
        with torch.cuda.stream(streams[j]):
            for _ in range(K):
                outs[j] = call(j)["traj_e"]

    ms = None
    for _ in range(2):  # warm-up round, then the timed one
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(j,)) for j in range(2)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (2 * K) * 1e3
    same = all(torch.equal(outs[j], ref[j]) for j in range(2))
    return {"value": Nq * T / (ms * 1e-3), "unit": "query-points*frames/s", "ms_per_clip": ms, "clips_in_flight": 2, "clips_timed": 2 * K,
            "results_identical_to_one_at_a_time": bool(same),
            "note": "two independent clips in flight on ONE GPU (two model objects / host threads / streams); a serving-layer throughput, "
                    "not the headline value: the reference forward is batch 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="BASELINE.json workload (default: the metric's config C3)")
    ap.add_argument("--views", type=int)
    ap.add_argument("--frames", type=int)
    ap.add_argument("--size", type=int)
    ap.add_argument("--queries", type=int, help="queries per GPU")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"],
                    help="matrix-core arithmetic of convs/linears (default: the config's dtype; BASELINE.json quotes C3 in bf16)")
    ap.add_argument("--cpu-hw", type=int, default=0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-corr-calibration", action="store_true",
                    help="skip the back-to-back calibration launches of the correlation kernel after the timed region (use under rocprofv3: "
                         "the roofline kernel's row of the kernel statistics then holds the in-situ launches only)")
    ap.add_argument("--clips-in-flight", type=int, default=1, choices=[1, 2],
                    help="2: AFTER the headline measurement, additionally time two independent clips in flight (two model objects, two "
                         "host threads, two streams) and print it as the separate field `throughput_two_clips` (never `value`: the "
                         "reference forward is batch 1)")
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

    V, T, H, W, Nq, prec, clip_kw, what = CONFIGS[args.config]
    V, T, Nq = args.views or V, args.frames or T, args.queries or Nq
    if args.size:
        H = W = args.size
    prec = args.precision or prec

    model = MVTracker(hidden_size=256).eval()
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev)
    model.precision = prec
    clip = synth.make_clip(1234, V=V, T=T, H=H, W=W, N=Nq * world, **clip_kw)  # same clip on every rank
    a = {k: torch.from_numpy(v).to(dev) for k, v in clip.items()}
    runner = ShardedTracker(model)

    # live timing with HIP events on the launch stream (torch events record on the current stream, which is the launch stream):
    # every launch of the correlation kernel, every updater call, every encoder chunk
    corr_ev, upd_ev, enc_ev = [], [], []
    timing = {"on": False}
    real_corr, real_upd, real_enc = hip.corr_gather_dot, model._update_former, model._encode

    last_call = {}

    def timed(real, sink, tag):
        def f(*cargs, **ckw):
            if real is real_corr:
                last_call["corr"] = (cargs, ckw)  # (keeps the operands of the most recent launch alive for the calibration below)
            if not timing["on"]:
                return real(*cargs, **ckw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = real(*cargs, **ckw)
            e1.record()
            sink.append((e0, e1, tag(*cargs, **ckw)))
            return r
        return f

    hip.corr_gather_dot = timed(real_corr, corr_ev, lambda *c, **k: c[7] * c[8])          # rows = N * S
    # (bf16 mode runs everything after the correlation -- token assembly, updater, flow head, track update -- as ONE library call)
    hip.updateformer_forward_tokens = timed(hip.updateformer_forward_tokens, upd_ev, lambda w, co, fc, Fc, ff, Cf, mv, po, te, E, n, *r, **k: n)
    model._update_former = timed(real_upd, upd_ev, lambda pk, x, ldx, n, *r, **k: n)       # tracks of the call
    model._encode = timed(real_enc, enc_ev, lambda pk, x4, n, *r, **k: n)                  # images of the chunk

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
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()  # (no sync between steps: the K steps are timed as one bracketed region)
    barrier()
    dt = time.perf_counter() - t0
    timing["on"] = False
    runner.check_finite_collective()  # (every rank joins: a rank raising alone would strand its peers)
    # Calibration of the HIP-event timing of the correlation kernel, outside the timed region, with the operands of the last launch
    # and nothing else on the chip: (a) the same launch 20 x back to back inside ONE event pair -> its duration alone, launch gaps
    # included, no event cost (the ~225 MB it touches may stay cache-resident between the repeats: a WARM figure); (b) 20 single
    # launches, one event pair each, exactly as they are timed in situ: (b) - (a) is what an event pair adds to a ~35 us kernel (the
    # marker packets' own latency); (c) single launches, each behind a 512 MiB write to another buffer (nothing of the store is left
    # in the 256 MiB Infinity Cache): the COLD figure alone on the chip.  `roofline.frac` is the in-situ measurement as it is; the
    # corrected and the alone figures are reported under their own keys.  --no-corr-calibration skips all of this (rocprofv3 runs).
    alone_b2b_ms = ev_overhead_ms = alone_cold_ms = None
    if rank == 0 and "corr" in last_call and not args.no_corr_calibration:
        cargs, ckw = last_call["corr"]
        torch.cuda.synchronize()
        for _ in range(5):
            real_corr(*cargs, **ckw)
        b2b, single, cold = [], [], []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                real_corr(*cargs, **ckw)
            e1.record()
            torch.cuda.synchronize()
            b2b.append(e0.elapsed_time(e1) / 20)
            pairs = []
            for _ in range(20):  # (enqueued back to back like the in-situ launches: the stream stays busy between the pairs)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                real_corr(*cargs, **ckw)
                e1.record()
                pairs.append((e0, e1))
            torch.cuda.synchronize()
            single += [e0.elapsed_time(e1) for e0, e1 in pairs]
        flush = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
        pairs = []
        for _ in range(12):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            real_corr(*cargs, **ckw)
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        cold = [e0.elapsed_time(e1) for e0, e1 in pairs[2:]]
        del flush
        alone_b2b_ms = float(np.median(b2b))
        ev_overhead_ms = max(0.0, float(np.median(single)) - alone_b2b_ms)
        alone_cold_ms = float(np.median(cold)) - ev_overhead_ms
    two_clips = None
    if rank == 0 and world == 1 and args.clips_in_flight == 2:
        two_clips = two_clips_in_flight(model, a, args, MVTracker, synth, dev, V, T, H, W, Nq, clip_kw, prec)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * Nq * T / (dt / args.steps)
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        store_bf16 = model.store_dtype() == torch.bfloat16
        full = [(e0.elapsed_time(e1), rows) for e0, e1, rows in corr_ev if rows == Nq * model.S]
        raw_ms = float(np.mean([m for m, _ in full])) if full else float("nan")
        ev_overhead_ms = ev_overhead_ms or 0.0
        kern_ms = raw_ms - ev_overhead_ms  # launch duration without the event pair's own cost (a derived figure: own key)
        alg = model.corr_n_levels * corr_algorithmic_bytes(Nq * model.S, model.corr_neighbors, model.latent_dim, 2 if store_bf16 else 4)
        achieved = alg / (raw_ms * 1e-3) / 1e9 if full else float("nan")  # as measured in situ, nothing subtracted
        # the launches of the LAST window run alone on the chip; the earlier windows share HBM with the encoder of the later frames
        # on the second stream.  per step: windows x iters full launches, in order
        per_win = args.iters
        n_full_step = len(full) // args.steps if args.steps else 0
        alone = [m for i, (m, _) in enumerate(full) if n_full_step and (i % n_full_step) >= n_full_step - per_win]
        last_win_ms = (float(np.mean(alone)) - ev_overhead_ms) if (alone and model.overlap_encoder and n_full_step > per_win) else None
        alone_ms = alone_b2b_ms if (alone_b2b_ms and last_call["corr"][0][7] * last_call["corr"][0][8] == Nq * model.S) else None
        # HBM traffic of the roofline kernel: an OFFLINE PMC measurement committed under profiles/ (tools/pmc_traffic.sh: counters
        # cannot be collected from inside the timed run).  Only quoted for the workload it was measured on; the source is named.
        traffic, traffic_source = None, None
        for name in ("r04_corr_traffic.json", "r03_corr_traffic.json", "r02_corr_traffic.json", "r01_corr_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("algorithmic_bytes_per_launch") == alg:
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_source = f"profiles/{name} (offline rocprofv3 PMC passes: FETCH_SIZE, WRITE_SIZE; not measured in this run)"
                    break
        peak_tf = MFMA_F32_TFLOPS if prec == "fp32" else MFMA_BF16_TFLOPS
        def union_ms(evs):
            """Wall time during which at least one of the timed calls was running: the encoder's chunks run on up to three
            streams at once, so the SUM of their event-pair times counts overlapped wall twice (event timestamps are global: every
            interval is placed on the axis of the first step mark)."""
            iv = sorted((marks[0].elapsed_time(e0), marks[0].elapsed_time(e1)) for e0, e1, _ in evs)
            tot, hi = 0.0, float("-inf")
            for a_, b_ in iv:
                if a_ > hi:
                    tot += b_ - a_
                    hi = b_
                elif b_ > hi:
                    tot += b_ - hi
                    hi = b_
            return tot

        upd_ms = union_ms(upd_ev) / args.steps
        upd_sum = sum(e0.elapsed_time(e1) for e0, e1, _ in upd_ev) / args.steps
        upd_fl = sum(updater_flops(n, model.S) for _, _, n in upd_ev) / args.steps
        enc_ms = union_ms(enc_ev) / args.steps
        enc_sum = sum(e0.elapsed_time(e1) for e0, e1, _ in enc_ev) / args.steps
        enc_fl = sum(n * encoder_flops(H, W, model.latent_dim) for _, _, n in enc_ev) / args.steps
        x3 = 3.0 if prec == "bf16x3" else 1.0  # three bf16 MFMAs per product in the split-precision mode

        def mfma(fl, t_ms, calls, t_sum):
            tf = fl / (t_ms * 1e-3) / 1e12 if t_ms > 0 else float("nan")
            return {"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf * x3 / peak_tf,
                    "flops_per_step": fl, "ms_per_step": t_ms, "ms_per_step_sum_of_calls": t_sum, "calls_per_step": calls,
                    "timing": "in-situ wall (union of the calls' HIP-event intervals over all streams)"}

        out = {
            # (BASELINE.json's metric string for its config C3; the other workloads are labelled by their own shape)
            "metric": "query-points*frames/sec, 4-view 24-frame 512x512 @1024 queries" if (args.config == "c3" and (V, T, H, W, Nq) == (4, 24, 512, 512, 1024))
                      else f"query-points*frames/sec, {V}-view {T}-frame {H}x{W} @{Nq} queries",
            "value": value, "unit": "query-points*frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "ms_per_step_median": float(np.median(per_step)), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (split-precision bf16 MFMA, f32 accumulate, f32-grade)", "bf16": "bf16"}[prec],
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {what}; {V}-view {T}-frame {H}x{W}, {Nq} queries per GPU, corr K=16 x 4 levels, "
                                   f"iters={args.iters}, {len(model.last_windows)} windows, random-init weights (seeded recipe)",
                       "queries_total": Nq * world, "parallelism": f"query-shard x{world}",
                       "frame_store": "bf16 rows" if store_bf16 else "fp32 rows"},
            "roofline": {"bound": "hbm", "kernel": "corr_gather_dot_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # `achieved` / `frac` price the ALGORITHMIC bytes over the average in-situ launch duration
                         # as the HIP events measured it (all windows: two of three share HBM with the encoder stream; the event
                         # pair's own cost included).  Derived figures, each under its own key: `frac_event_corrected`: the measured
                         # event-pair overhead subtracted; `frac_alone_cold`: one launch alone on the chip behind a 512 MiB cache
                         # flush; `frac_alone_warm`: 20 x back to back in one event pair (operands may stay cache-resident);
                         # `frac_last_window`: the in-situ launches of the last window (no encoder stream), overhead subtracted
                         "frac_event_corrected": (alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (full and ev_overhead_ms) else None,
                         "frac_alone_cold": (alg / (alone_cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (alone_ms and alone_cold_ms) else None,
                         "frac_alone_warm": (alg / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if alone_ms else None,
                         "frac_last_window": (alg / (last_win_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if last_win_ms else None,
                         "hbm_GBps_from_traffic": (traffic / (raw_ms * 1e-3) / 1e9) if (traffic and full) else None,
                         "algorithmic_bytes_per_launch": alg, "bytes_per_unit": alg // (model.corr_n_levels * Nq * model.S),
                         "avg_launch_ms": raw_ms, "avg_launch_ms_event_corrected": kern_ms, "event_pair_overhead_ms": ev_overhead_ms,
                         "avg_launch_ms_alone_warm": alone_ms, "avg_launch_ms_alone_cold": alone_cold_ms if alone_ms else None,
                         "launches_timed": len(full), "min_launch_ms": float(np.min([m for m, _ in full])) if full else None},
            "roofline_mfma": {"updater": mfma(upd_fl, upd_ms, len(upd_ev) // args.steps, upd_sum),
                              "encoder": mfma(enc_fl, enc_ms, len(enc_ev) // args.steps, enc_sum)},
        }
        if two_clips is not None:
            out["throughput_two_clips"] = two_clips
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, V, (H, W))
        print(json.dumps(out), flush=True)
    if world > 1 or os.environ.get("MVT_FORCE_SHARDED"):
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
