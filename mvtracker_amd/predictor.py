"""EvaluationPredictor on MI355X: the drop-in boundary of the tracking forward path.

Mirrors ``mvtracker.models.evaluation_predictor_3dpt.EvaluationPredictor`` (reference
evaluation_predictor_3dpt.py:17-414): same constructor, same ``forward`` keyword arguments
(``rgbs, depths, query_points_3d, intrs, extrs`` + ignored ``**kwargs``), same result dict
(``traj_e``, ``vis_e`` bool, ``vis_e_as_prob``), so ``demo.py`` and ``Evaluator.evaluate_sequence``
can drive it unchanged.  The heavy lifting is ``self.model`` (mvtracker_amd.tracker.MVTracker);
the resize runs in the HIP library, the support-grid synthesis is a few hundred points of
bookkeeping on device tensors.
"""
from __future__ import annotations

import contextlib
import logging
from typing import Optional, Tuple

import torch

from . import hip

log = logging.getLogger(__name__)


def _bilinear_sample_depth(depth: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Four clamped taps with weights from the unclamped corners (reference model_utils.py:81-165).

    depth (H,W), x,y (M,) pixel coordinates -> (M,)."""
    H, W = depth.shape
    x0, y0 = torch.floor(x), torch.floor(y)
    x1, y1 = x0 + 1, y0 + 1
    cx0, cx1 = x0.clamp(0, W - 1).long(), x1.clamp(0, W - 1).long()
    cy0, cy1 = y0.clamp(0, H - 1).long(), y1.clamp(0, H - 1).long()
    flat = depth.reshape(-1)
    return ((x1 - x) * (y1 - y) * flat[cy0 * W + cx0] + (x - x0) * (y1 - y) * flat[cy0 * W + cx1]
            + (x1 - x) * (y - y0) * flat[cy1 * W + cx0] + (x - x0) * (y - y0) * flat[cy1 * W + cx1])


def _bilinear_sample_depth_multi(depths: torch.Tensor, v: torch.Tensor, t: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``_bilinear_sample_depth`` for points that live in different (view, frame) depth maps: depths (V,T,H,W), v,t,x,y (M,)."""
    V, T, H, W = depths.shape
    x0, y0 = torch.floor(x), torch.floor(y)
    x1, y1 = x0 + 1, y0 + 1
    cx0, cx1 = x0.clamp(0, W - 1).long(), x1.clamp(0, W - 1).long()
    cy0, cy1 = y0.clamp(0, H - 1).long(), y1.clamp(0, H - 1).long()
    flat = depths.reshape(-1)
    base = (v * T + t) * (H * W)
    return ((x1 - x) * (y1 - y) * flat[base + cy0 * W + cx0] + (x - x0) * (y1 - y) * flat[base + cy0 * W + cx1]
            + (x1 - x) * (y - y0) * flat[base + cy1 * W + cx0] + (x - x0) * (y - y0) * flat[base + cy1 * W + cx1])


def get_uniformly_sampled_pts(size: int, num_frames: int, extent: Tuple[float, ...], device="cpu") -> torch.Tensor:
    """(1, size, 3) rows (t, a, b): t uniform in [0, num_frames), a uniform in [0, extent[1]), b in [0, extent[0]) -- reference
    evaluation_predictor_3dpt.py:417-429, the same draws in the same order (torch.randint, then torch.rand)."""
    time_points = torch.randint(low=0, high=num_frames, size=(size, 1), device=device)
    space_points = torch.rand(size, 2, device=device) * torch.tensor([extent[1], extent[0]], device=device)
    return torch.cat((time_points, space_points), dim=1)[None]


def points_on_a_grid(size: int, extent: Tuple[float, float], center=None, device="cpu") -> torch.Tensor:
    """(size*size, 2) pixel (x, y) grid with margin W/64 (reference model_utils.py:361-417)."""
    if size == 1:
        return torch.tensor([[extent[1] / 2, extent[0] / 2]], device=device)
    if center is None:
        center = [extent[0] / 2, extent[1] / 2]
    m = extent[1] / 64
    ys = torch.linspace(m - extent[0] / 2 + center[0], extent[0] / 2 + center[0] - m, size, device=device)
    xs = torch.linspace(m - extent[1] / 2 + center[1], extent[1] / 2 + center[1] - m, size, device=device)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1).reshape(-1, 2)


class EvaluationPredictor(torch.nn.Module):
    def __init__(
            self,
            multiview_model: torch.nn.Module,
            interp_shape: Optional[Tuple[int, int]] = (384, 512),
            visibility_threshold=0.5,
            grid_size: int = 5,
            n_grids_per_view: int = 1,
            local_grid_size: int = 8,
            local_extent: int = 50,
            single_point: bool = False,
            sift_size: int = 0,
            num_uniformly_sampled_pts: int = 0,
            n_iters: int = 6,
    ) -> None:
        super().__init__()
        self.model = multiview_model
        self.interp_shape = interp_shape
        self.visibility_threshold = visibility_threshold
        self.grid_size = grid_size
        self.n_grids_per_view = n_grids_per_view
        self.local_grid_size = local_grid_size
        self.local_extent = local_extent
        self.single_point = single_point
        self.sift_size = sift_size
        self.num_uniformly_sampled_pts = num_uniformly_sampled_pts
        self.n_iters = n_iters
        self.single_point_streams = 8  # HIP streams the per-query forwards of single_point mode are spread over
        self._stream_pool = {}
        self.model.eval()

    # ---- helpers -------------------------------------------------------------------------
    def _streams(self, dev, n):
        key = (dev.type, dev.index)
        pool = self._stream_pool.setdefault(key, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=dev))
        return pool[:n]

    @staticmethod
    def _invert(intrs, extrs):
        """K^-1 (V,T,3,3) and rows 0..2 of [R|t]^-1 (V,T,3,4) through the library (fp64 closed form)."""
        V, T = intrs.shape[:2]
        kinv = torch.empty(V * T, 9, device=intrs.device)
        einv = torch.empty(V * T, 12, device=intrs.device)
        hip.invert_cameras(intrs.reshape(V * T, 9).contiguous(), extrs.reshape(V * T, 12).contiguous(), kinv, einv, V * T)
        return kinv.reshape(V, T, 3, 3), einv.reshape(V, T, 3, 4)

    @staticmethod
    def _unproject(pix, z, kinv, einv):
        ph = torch.cat([pix, torch.ones_like(pix[:, :1])], 1)
        cam = (ph @ kinv.t()) * z[:, None]
        return cam @ einv[:, :3].t() + einv[:, 3]

    def _support_rows(self, depth, pix, kinv, einv, t):
        z = _bilinear_sample_depth(depth, pix[:, 0], pix[:, 1])
        world = self._unproject(pix, z, kinv, einv)
        return torch.cat([torch.full_like(world[:, :1], float(t)), world], 1)

    # ---- forward ---------------------------------------------------------------------------
    @torch.no_grad()
    @hip.guarded
    def forward(
            self,
            rgbs,
            depths,
            query_points_3d,
            intrs,
            extrs,
            save_debug_logs=False,
            debug_logs_path="",
            query_points_view=None,
            **kwargs,
    ):
        batch_size, num_views, num_frames, _, height_raw, width_raw = rgbs.shape
        _, num_points, _ = query_points_3d.shape
        assert rgbs.shape == (batch_size, num_views, num_frames, 3, height_raw, width_raw)
        assert depths.shape == (batch_size, num_views, num_frames, 1, height_raw, width_raw)
        assert query_points_3d.shape == (batch_size, num_points, 4)
        assert intrs.shape == (batch_size, num_views, num_frames, 3, 3)
        assert extrs.shape == (batch_size, num_views, num_frames, 3, 4)
        if batch_size != 1:
            raise NotImplementedError
        if self.sift_size > 0:
            raise NotImplementedError
        hip.require_device(rgbs)
        dev = rgbs.device
        V, T = num_views, num_frames
        # (uint8 frames stay uint8 unless they have to be resized: the encoder's first kernel converts them)
        if not (rgbs.dtype == torch.uint8 and self.interp_shape is None):
            rgbs = rgbs.to(torch.float32)
        rgbs = rgbs.contiguous()
        depths = depths.to(torch.float32).contiguous()
        intrs = intrs.to(torch.float32)
        extrs = extrs.to(torch.float32)
        query_points_3d = query_points_3d.to(torch.float32)

        if self.interp_shape is None:  # evaluation_predictor_3dpt.py:72-87
            height, width = height_raw, width_raw
        else:
            height, width = self.interp_shape
            r = torch.empty(1, V, T, 3, height, width, device=dev)
            hip.resize_nearest(rgbs, r, V * T * 3, height_raw, width_raw, height, width)
            d = torch.empty(1, V, T, 1, height, width, device=dev)
            hip.resize_nearest(depths, d, V * T, height_raw, width_raw, height, width)
            rgbs, depths = r, d
            rs = torch.tensor([[width / width_raw, 0, 0], [0, height / height_raw, 0], [0, 0, 1]], device=dev,
                              dtype=intrs.dtype)
            intrs = torch.einsum("ij,BVTjk->BVTik", rs, intrs)

        kinv, einv = self._invert(intrs[0], extrs[0])
        support = torch.zeros(0, 4, device=dev)
        if self.grid_size > 0:  # :101-120
            pix = points_on_a_grid(self.grid_size, (height, width), device=dev)
            rows = []
            for t in range(0, T, max(1, T // self.n_grids_per_view)):
                for v in range(V):
                    rows.append(self._support_rows(depths[0, v, t, 0], pix, kinv[v, t], einv[v, t], t))
            support = torch.cat(rows, 0)

        if self.num_uniformly_sampled_pts > 0:  # :147-190
            # Uniformly sampled support points: (t, y, x) rows from get_uniformly_sampled_pts -- the same two torch draws as the
            # reference, from the global generator of the tensors' device -- lifted into the world through EVERY view's depth map
            # (sample-major, then view), with the reference's argument order kept as it is: the column it calls y is scaled by the
            # WIDTH and the one it calls x by the HEIGHT (:424-426), and out-of-map coordinates are clamped by the sampler.
            sp = get_uniformly_sampled_pts(self.num_uniformly_sampled_pts, T, (height, width), device=dev)[0]
            t_s, y_s, x_s = sp[:, 0].long(), sp[:, 1].float(), sp[:, 2].float()
            vid = torch.arange(V, device=dev).repeat(sp.shape[0])
            fid, xs, ys = t_s.repeat_interleave(V), x_s.repeat_interleave(V), y_s.repeat_interleave(V)
            z = _bilinear_sample_depth_multi(depths[0, :, :, 0], vid, fid, xs, ys)
            camm = torch.einsum("mij,mj->mi", kinv[vid, fid], torch.stack([xs, ys, torch.ones_like(xs)], 1)) * z[:, None]
            ei = einv[vid, fid]
            world = torch.einsum("mij,mj->mi", ei[:, :, :3], camm) + ei[:, :, 3]
            support = torch.cat([support, torch.cat([fid[:, None].to(world.dtype), world], 1)], 0)

        nan_flags = []
        fwd = dict(intrs=intrs, extrs=extrs, iters=self.n_iters, save_debug_logs=save_debug_logs,
                   debug_logs_path=debug_logs_path, query_points_view=query_points_view, **kwargs)
        if self.single_point:  # :191-339, one forward per query with its local grids
            traj_e = torch.zeros(1, T, num_points, 3, device=dev)
            vis_e = torch.zeros(1, T, num_points, device=dev)
            qt_d = query_points_3d[0, :, 0].long()
            qt = qt_d.cpu().tolist()
            # The N forwards differ only in their queries: encoder, feature pyramid and point clouds are built ONCE and
            # shared (the reference re-encodes the clip for every query).  Every forward reads
            # frames >= its own first query frame only, so one store from the earliest query frame serves them all.
            if hasattr(self.model, "build_frame_store"):
                f32 = lambda t_: t_.to(torch.float32).contiguous()
                t_first = 0 if (self.grid_size > 0 or not qt) else max(0, min(qt))  # (the global support grid starts at frame 0)
                r0 = rgbs[0].contiguous() if rgbs.dtype == torch.uint8 else f32(rgbs[0])
                fwd["frame_store"] = self.model.build_frame_store(r0, f32(depths[0]), f32(intrs[0]), f32(extrs[0]),
                                                                  t0=t_first)
            # Local support grids of ALL queries in one pass (:221-252): project every query into every view at its own frame
            # (one device op, ONE readback instead of two .item() syncs per view and query), lay the grids out and clip them to
            # the image on the host, then sample depth / unproject all surviving pixels at once on the device.
            local_rows = [[] for _ in range(num_points)]
            if self.local_grid_size > 0 and num_points > 0:
                qh = torch.cat([query_points_3d[0, :, 1:], torch.ones(num_points, 1, device=dev)], 1)
                cam = torch.einsum("vnij,nj->vni", extrs[0][:, qt_d], qh)
                ph = torch.einsum("vnij,vnj->vni", intrs[0][:, qt_d], cam)
                centres = (ph[..., :2] / ph[..., 2:]).cpu()  # (V, N, 2) pixel (x, y)
                pix_l, vid_l, fid_l, counts = [], [], [], []
                for i in range(num_points):
                    for v in range(V):
                        px, py = centres[v, i, 0].item(), centres[v, i, 1].item()
                        pix = points_on_a_grid(self.local_grid_size, (self.local_extent, self.local_extent), (py, px), "cpu")
                        ok = (pix[:, 0] >= 0) & (pix[:, 0] < width) & (pix[:, 1] >= 0) & (pix[:, 1] < height)
                        k = int(ok.sum())
                        counts.append(k)
                        if k:
                            pix_l.append(pix[ok])
                            vid_l.append(torch.full((k,), v, dtype=torch.long))
                            fid_l.append(torch.full((k,), qt[i], dtype=torch.long))
                if pix_l:
                    pix_all = torch.cat(pix_l).to(dev)
                    vid, fid = torch.cat(vid_l).to(dev), torch.cat(fid_l).to(dev)
                    z = _bilinear_sample_depth_multi(depths[0, :, :, 0], vid, fid, pix_all[:, 0], pix_all[:, 1])
                    phm = torch.cat([pix_all, torch.ones_like(pix_all[:, :1])], 1)
                    camm = torch.einsum("mij,mj->mi", kinv[vid, fid], phm) * z[:, None]
                    ei = einv[vid, fid]
                    world = torch.einsum("mij,mj->mi", ei[:, :, :3], camm) + ei[:, :, 3]
                    rows_all = torch.cat([fid[:, None].to(world.dtype), world], 1)
                    o = 0
                    for i in range(num_points):
                        for v in range(V):
                            k = counts[i * V + v]
                            if k:
                                local_rows[i].append(rows_all[o:o + k])
                                o += k
            # The per-query forwards are independent (each has its own 64 virtual tracks, :254-275) and tiny -- a few hundred
            # tracks, pure launch-latency chains -- so they are issued round-robin on a few HIP streams and overlap on the GPU.
            n_streams = max(1, min(self.single_point_streams, num_points)) if dev.type == "cuda" else 1
            main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
            streams = self._streams(dev, n_streams) if n_streams > 1 else []
            for s_ in streams:
                s_.wait_stream(main)
            for i in range(num_points):
                ctx = torch.cuda.stream(streams[i % n_streams]) if streams else contextlib.nullcontext()
                with ctx:  # (everything of query i, its input rows included, is enqueued on its own stream)
                    q_i = torch.cat([query_points_3d[0, i:i + 1]] + local_rows[i] + [support], 0)[None]
                    res = self.model(rgbs, depths=depths, query_points=q_i, **fwd)
                    traj_e[:, :, i] = res["traj_e"][:, :, 0]
                    vis_e[:, :, i] = res["vis_e"][:, :, 0]
                    nan_flags.append(getattr(self.model, "last_nan_flag", None))
            for s_ in streams:
                main.wait_stream(s_)
        else:  # joint mode, :341-360
            q = torch.cat([query_points_3d[0], support], 0)[None]
            res = self.model(rgbs, depths=depths, query_points=q, **fwd)
            traj_e = res["traj_e"][:, :, :num_points, :]
            vis_e = res["vis_e"][:, :, :num_points]
            nan_flags.append(getattr(self.model, "last_nan_flag", None))
        # deferred NaN guard of the model (reference mvtracker.py:401-404 logs in the iteration where the NaN appears): the
        # caller is about to copy the tracks to the host anyway, so one flag read here costs nothing
        flags = [f for f in nan_flags if f is not None]
        if flags and int(torch.stack([f.reshape(()) for f in flags]).max().item()) != 0:
            log.error("Got NaN values in coords, perhaps the training exploded")
            self.last_nan = True
        else:
            self.last_nan = False
        return {"traj_e": traj_e, "vis_e": vis_e > self.visibility_threshold, "vis_e_as_prob": vis_e}
