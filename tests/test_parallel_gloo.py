"""N > 1 path on CPU: two `gloo` ranks run mvtracker_amd.parallel.ShardedTracker (kernels replaced by tests/hip_mock.py).
Checks the frame-split encode + all-gather, the query partition and the output gather against the oracle run on
each shard separately (shard = independent forward)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CASES = {
    # name: (world, clip kwargs, overlap_encoder)
    "two_ranks": (2, dict(seed=71, V=2, T=12, H=128, W=128, N=9), True),
    # 3 ranks, 2 views x 15 frames: the first block has 24 images (8 per rank), the second 6 (2 per rank); late queries, 2 windows
    "three_ranks_late": (3, dict(seed=72, V=2, T=15, H=96, W=96, N=10, late_queries=True, query_frames=(2, 5)), True),
    # 4 ranks, 1 view x 14 frames: second block = 2 images over 4 ranks -> ranks 2 and 3 have EMPTY shares (zeros travel);
    # the first block (12 images, 3 per rank) is even.  Same clip once more as a single 14-image block (4,4,4,2: a short share
    # plus 2 images of spill into the tail padding).
    "four_ranks_empty_share": (4, dict(seed=73, V=1, T=14, H=128, W=128, N=8), True),
    "four_ranks_one_block": (4, dict(seed=73, V=1, T=14, H=128, W=128, N=8), False),
    # 8 ranks (the node size), 1 view x 14 frames: block one = 12 images, 2 per rank -> ranks 6 and 7 EMPTY; block two = 2 images,
    # 1 per rank -> ranks 2..7 EMPTY: several ranks have empty shares in BOTH blocks; one query per rank
    "eight_ranks_empty_shares": (8, dict(seed=74, V=1, T=14, H=128, W=128, N=8), True),
    # the in-place form of the exchange (MVT_GATHER_INPLACE=1: the input aliases its slot of the output), uneven shares + spill;
    # every other case runs the default staged form
    "three_ranks_inplace": (3, dict(seed=72, V=2, T=15, H=96, W=96, N=10, late_queries=True, query_frames=(2, 5)), "inplace"),
}


def _worker(rank, world, port, out_path, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2 if world <= 4 else 1)
    if CASES[case][2] == "inplace":
        os.environ["MVT_GATHER_INPLACE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hip_mock
    from mvtracker_amd import hip, synth
    from mvtracker_amd.parallel import ShardedTracker
    from mvtracker_amd.tracker import MVTracker

    class MP:
        def setattr(self, obj, name, val):
            setattr(obj, name, val)

    hip_mock.install(MP())
    m = MVTracker(hidden_size=256).eval()
    sd = synth.make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    _, kw, overlap = CASES[case]
    m.overlap_encoder = bool(overlap)
    clip = synth.make_clip(**kw)
    a = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    st = ShardedTracker(m)
    assert st.staged == (overlap != "inplace")
    res = st(*a, iters=2)
    # the exchanged level-0 store (image-granular shares, spill into the next block / the tail padding, no staging copy) must
    # equal a single-rank encode of the same frames, image for image, on EVERY rank
    t0 = int(a[2][0, :, 0].long().min())
    whole = m.encode_frames(a[0][0].float(), t0, kw["T"])
    # (the mock's CPU convolutions round differently for different batch sizes: a tolerance far below any layout error, which
    #  would put a different IMAGE -- or zeros -- into a slot)
    diff = (st.last_store["fvec"][0][t0:].float() - whole[t0:].float()).abs().amax(dim=(2, 3, 4))
    assert float(diff.max()) < 1e-4 * float(whole[t0:].abs().max()), f"rank {rank}: exchanged store differs from the single-rank encode: {diff}"
    if rank == 0:
        np.savez(out_path, traj=res["traj_e"].numpy(), vis=res["vis_e"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", list(CASES))
def test_sharded_tracker(tmp_path, case):
    world, kw, _ = CASES[case]
    out = str(tmp_path / "out.npz")
    port = 29500 + (os.getpid() * 7 + len(case)) % 2000
    mp.spawn(_worker, args=(world, port, out, case), nprocs=world, join=True)
    got = np.load(out)
    from mvtracker_amd import synth
    from mvtracker_amd.parallel import ShardedTracker
    from oracle import mvt_oracle as O
    cfg = O.TrackerConfig()
    W = O.make_weights(cfg, 0)
    clip = synth.make_clip(**kw)
    a = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    N, T = kw["N"], kw["T"]
    assert got["traj"].shape == (1, T, N, 3)
    for rank in range(world):
        lo, hi = ShardedTracker.shard_bounds(N, world, rank)
        ro = O.tracker_forward(W, cfg, a[0], a[1], a[2][:, lo:hi], a[3], a[4], iters=2, knn_mode="exact")
        # every shard starts its windows at ITS earliest query; frames before that stay zero on both sides
        ref = ro["traj_e"].numpy()
        assert np.abs(got["traj"][:, :, lo:hi] - ref).max() / np.abs(ref).max() < 1e-4, (case, rank)
        assert np.abs(got["vis"][:, :, lo:hi] - ro["vis_e"].numpy()).max() < 2e-3, (case, rank)


def test_shard_bounds_cover_everything():
    from mvtracker_amd.parallel import ShardedTracker
    for n in (1, 7, 8, 9, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [ShardedTracker.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
