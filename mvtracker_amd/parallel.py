"""Multi-GPU execution of the tracking forward path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Contract: the query set is partitioned into contiguous shards and every
shard is an INDEPENDENT forward -- exactly what the reference does when it runs query subsets
(evaluation_predictor_3dpt.py:191-277, demo.py:809-824); the 64 virtual tracks of the updater couple
the tracks of one call, so a shard's result equals the reference run on that subset, not on the
union.  The only data-path exchange happens once, before the refinement loop: the V x T images are cut
evenly across the ranks, each rank encodes its run of images straight into its slot of the level-0 store and the slots are
all-gathered in place (each rank then derives pyramid levels 1-3 and the point clouds locally).  No collective runs inside
the refinement loop; the small outputs are all-gathered at the end when requested.
"""
from __future__ import annotations

import logging
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import hip

log = logging.getLogger(__name__)


class ShardedTracker:
    def __init__(self, model, group: Optional["dist.ProcessGroup"] = None):
        self.model = model
        self.group = group
        # The exchange stages this rank's share (a copy of 1 / world of the block: 50 MB of the 403 MB of a C3 clip at 8 GPUs, ~10 us)
        # unless MVT_GATHER_INPLACE=1 asks for the aliasing in-place form.  The staged form is the DEFAULT because the multi-GPU
        # path has never run on hardware at world > 1: an input that aliases the output is what
        # ncclAllGather documents as in-place, but nothing here has been able to verify that torch + RCCL accept it.
        self.staged = os.environ.get("MVT_GATHER_INPLACE", "0") == "0"
        self.last_store = None  # the frame store of the last call (tests compare it with a single-rank encode)

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    @staticmethod
    def shard_bounds(n: int, world: int, rank: int):
        per = (n + world - 1) // world
        return min(n, rank * per), min(n, (rank + 1) * per)

    def _all_gather(self, local: torch.Tensor, world: int) -> torch.Tensor:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        else:
            parts = list(out.chunk(world, dim=0))
            dist.all_gather(parts, local.contiguous(), group=self.group)
        return out

    def _all_gather_in_place(self, out: torch.Tensor, rank: int, world: int) -> None:
        """``out`` = world equal chunks along dim 0; this rank's chunk already holds its contribution.

        Which form runs is decided identically on every rank BEFORE the collective (a per-rank try / except around a collective
        can leave ranks in different code paths): a staged copy of the local share by default, or -- ``MVT_GATHER_INPLACE=1``, the
        same on all ranks because the launcher exports one environment -- the in-place form, where the input aliases this rank's
        slot of the output, the layout ncclAllGather documents as in-place (sendbuff == recvbuff + rank * sendcount)."""
        per = out.shape[0] // world
        mine = out[rank * per:(rank + 1) * per]
        src = mine.clone() if self.staged else mine
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, src, group=self.group)
        else:  # gloo (CPU tests): the list form; c10d stages the input itself, so the aliasing input is safe
            dist.all_gather(list(out.chunk(world, dim=0)), src, group=self.group)

    @staticmethod
    def image_share(n_img: int, world: int, rank: int):
        """Cut of a block of ``n_img`` images across the ranks: (per, lo, hi) -- every rank owns a chunk of ``per`` image slots
        starting at rank * per, of which [lo, hi) (block-relative) are real images; the rest of a short or empty share is zero
        filled.  world * per - n_img <= world - 1 slots spill past the block."""
        per = (n_img + world - 1) // world
        return per, min(n_img, rank * per), min(n_img, (rank + 1) * per)

    @torch.no_grad()
    @hip.guarded
    def __call__(self, rgbs, depths, query_points, intrs, extrs, iters=4, gather_output=True):
        world, rank = self._world()
        m = self.model
        if world == 1 and not (os.environ.get("MVT_FORCE_SHARDED") and dist.is_available() and dist.is_initialized()):
            return m(rgbs, depths, query_points, intrs, extrs, iters=iters)
        _, V, T, _, H, W = rgbs.shape
        N = query_points.shape[1]
        a, b = self.shard_bounds(N, world, rank)
        assert b > a, "fewer queries than ranks"
        t0 = int(query_points[0, :, 0].long().min().item())  # global first frame: identical on every rank
        store = None
        if t0 < T - m.S // 2:
            f32 = lambda t: t.to(torch.float32).contiguous()
            r0 = rgbs[0].contiguous() if rgbs.dtype == torch.uint8 else f32(rgbs[0])
            d0, i0, e0 = f32(depths[0]), f32(intrs[0]), f32(extrs[0])
            hs, ws, C = H // m.stride, W // m.stride, m.latent_dim
            dev = rgbs.device
            # Level-0 store [T][V][hs][ws][C] plus (world - 1) images of tail padding: an all-gather moves world equal
            # chunks, so a block whose image count is not a multiple of world spills at most world - 1 images past its end --
            # into the next block's frames (rewritten when that block is exchanged, later on the same or on the ordered side
            # stream) or, for the last block, into the padding.
            storage = torch.empty(T * V + world - 1, hs, ws, C, device=dev, dtype=m.store_dtype())
            level0 = storage[:T * V].view(T, V, hs, ws, C)
            level0[:t0].zero_()  # (never read: no window starts before the first query frame)

            def exchange_block(f0, f1):
                """Frames [f0, f1): the V * (f1 - f0) IMAGES are cut evenly across the ranks (a frame-granular split leaves
                ranks idle as soon as the block has fewer frames than ranks: 12 frames on 8 GPUs); every rank encodes its
                run of images straight into its chunk of the store and the chunks are all-gathered IN PLACE."""
                n_img = (f1 - f0) * V
                per, lo, hi = self.image_share(n_img, world, rank)
                base = f0 * V
                dst = storage[base:base + world * per]
                if hi > lo:
                    m.encode_images(r0, base + lo, base + hi, storage)
                if hi - lo < per:  # the unused rows of a short (or empty) share: zeros travel, never uninitialised memory
                    storage[base + rank * per + (hi - lo):base + (rank + 1) * per].zero_()
                self._all_gather_in_place(dst, rank, world)

            # The first window only reads frames [t0, t0+S): exchange those first and let the rest of the clip be encoded
            # and all-gathered on a second stream while the first windows are refined (events order each window after
            # the frames it reads, exactly like the single-GPU encoder overlap).
            first_end = min(T, t0 + m.S)
            two_blocks = m.overlap_encoder and first_end < T
            exchange_block(t0, first_end if two_blocks else T)
            store = m.build_frame_store(r0, d0, i0, e0, t0=t0, level0=level0, t1=first_end if two_blocks else T)
            if two_blocks and dev.type == "cuda":
                main = torch.cuda.current_stream(dev)
                side = m._side_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    # (this block is encoded BESIDE the first windows: short-lived workgroups only, MVT_IO_SHORT_WG -- bit-identical
                    #  results; see MVTracker.wide_conv_shared)
                    m._shared_gpu = getattr(m, "wide_conv_shared", "auto") != "1"
                    try:
                        exchange_block(first_end, T)
                    finally:
                        m._shared_gpu = False
                    m.fill_frame_features(store, r0, first_end, T, level0=level0)
                    ev = torch.cuda.Event()
                    ev.record(side)
                store["pending"] = [(first_end, ev)]
                for t_ in (r0, storage):  # allocated on the main stream, last used on the side stream
                    t_.record_stream(side)
            elif two_blocks:  # host tensors (the gloo tests): the same two exchanges, in order
                exchange_block(first_end, T)
                m.fill_frame_features(store, r0, first_end, T, level0=level0)
        self.last_store = store
        res = m(rgbs, depths, query_points[:, a:b], intrs, extrs, iters=iters, frame_store=store)
        if not gather_output:
            # (the caller owns the deferred NaN check -- and with more than one rank it must be check_finite_collective(): a rank
            #  that raised on model.check_finite() alone would leave its peers blocked in their next collective)
            return res
        per = (N + world - 1) // world
        traj = torch.zeros(per, T, 3, device=rgbs.device)
        vis = torch.zeros(per, T, device=rgbs.device)
        traj[:b - a] = res["traj_e"][0].permute(1, 0, 2)
        vis[:b - a] = res["vis_e"][0].t()
        traj = self._all_gather(traj, world)[:N].permute(1, 0, 2)[None]
        vis = self._all_gather(vis, world)[:N].t()[None]
        # The NaN guard (mvtracker.py:401-404) is a COLLECTIVE decision: a rank that raised on its own shard's flag would leave
        # its peers blocked in the next collective.  Every rank joins every collective of the call first, the per-shard flags
        # are max-reduced, and then all ranks raise (or none does) together -- one host read per call, after the last collective.
        self.check_finite_collective()
        return {"traj_e": traj, "vis_e": vis, "feat_init": res["feat_init"]}

    def check_finite_collective(self):
        """The deferred NaN guard of the last call as a COLLECTIVE decision (the only safe check after ``gather_output=False`` at
        world > 1): the per-shard flags are max-reduced and then every rank raises, or none does.  Every rank must call it."""
        world, _ = self._world()
        flag = getattr(self.model, "last_nan_flag", None)
        if flag is None:
            return
        if world > 1 or (dist.is_available() and dist.is_initialized()):
            flag = flag.clone()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()) != 0:
            raise FloatingPointError("Got NaN values in coords (on at least one query shard), perhaps the training exploded")
