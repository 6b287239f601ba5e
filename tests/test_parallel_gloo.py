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


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
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
    clip = synth.make_clip(71, V=2, T=12, H=128, W=128, N=9)
    a = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    res = ShardedTracker(m)(*a, iters=2)
    if rank == 0:
        np.savez(out_path, traj=res["traj_e"].numpy(), vis=res["vis_e"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_tracker_two_ranks(tmp_path):
    out = str(tmp_path / "out.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    from mvtracker_amd import synth
    from mvtracker_amd.parallel import ShardedTracker
    from oracle import mvt_oracle as O
    cfg = O.TrackerConfig()
    W = O.make_weights(cfg, 0)
    clip = synth.make_clip(71, V=2, T=12, H=128, W=128, N=9)
    a = [torch.from_numpy(clip[k]) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
    assert got["traj"].shape == (1, 12, 9, 3)
    for rank in range(2):
        lo, hi = ShardedTracker.shard_bounds(9, 2, rank)
        ro = O.tracker_forward(W, cfg, a[0], a[1], a[2][:, lo:hi], a[3], a[4], iters=2, knn_mode="exact")
        ref = ro["traj_e"].numpy()
        assert np.abs(got["traj"][:, :, lo:hi] - ref).max() / np.abs(ref).max() < 1e-4
        assert np.abs(got["vis"][:, :, lo:hi] - ro["vis_e"].numpy()).max() < 2e-3


def test_shard_bounds_cover_everything():
    from mvtracker_amd.parallel import ShardedTracker
    for n in (1, 7, 8, 9, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [ShardedTracker.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
