"""View assignment of the reference's MonocularToMultiViewAdapter (monocular_baselines.py:604-680).

The adapter runs a 2-D tracker per view; each 3-D query is handed to the single view in which it is best visible.  Only
that assignment rule -- the "integer view indices" of the parity clause -- is in scope; the
wrapped 2-D trackers are third-party models fetched from remote hubs."""
from __future__ import annotations

import torch

from . import hip


@torch.no_grad()
@hip.guarded
def assign_views(depths: torch.Tensor, query_points: torch.Tensor, intrs: torch.Tensor, extrs: torch.Tensor, return_projections=False):
    """depths (1,V,T,1,H,W), query_points (1,N,4), intrs (1,V,T,3,3), extrs (1,V,T,3,4) -> best view per query (1,N) int64
    [, pixel xy (1,V,N,2), camera z (1,V,N,1)] -- `query_points_best_visibility_view` of the reference."""
    B, V, T, _, H, W = depths.shape
    assert B == 1, "Batch size > 1 is not supported yet"
    N = query_points.shape[1]
    hip.require_device(depths)
    f32 = lambda t: t.to(torch.float32).contiguous()
    view = torch.empty(N, device=depths.device, dtype=torch.int32)
    xyz = torch.empty(V, N, 3, device=depths.device) if return_projections else None
    hip.adapter_best_view(f32(depths[0, :, :, 0]), f32(intrs[0]), f32(extrs[0]), f32(query_points[0]), V, T, H, W, N, view, xyz)
    if return_projections:
        return view.long()[None], xyz[None, :, :, :2], xyz[None, :, :, 2:]
    return view.long()[None]
