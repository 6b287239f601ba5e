"""Small geometry helpers of the evaluation path, device tensors in, device tensors out.

``world_to_pixel`` is the reference's ``world_space_to_pixel_xy_and_camera_z`` (model_utils.py:344-358), which the evaluator uses
to turn the 3-D tracks into the per-view 2-D tracks of the parity clause (evaluator_3dpt.py:554-561)."""
from __future__ import annotations

from typing import Tuple

import torch


@torch.no_grad()
def world_to_pixel(world_xyz: torch.Tensor, intrs: torch.Tensor, extrs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """world_xyz (T,N,3), intrs (T,3,3), extrs (T,3,4) -> pixel xy (T,N,2), camera z (T,N,1)."""
    assert world_xyz.ndim == 3 and world_xyz.shape[-1] == 3 and intrs.shape[-2:] == (3, 3) and extrs.shape[-2:] == (3, 4)
    wh = torch.cat([world_xyz, torch.ones_like(world_xyz[..., :1])], -1)
    cam = torch.einsum("Aij,ABj->ABi", extrs.to(world_xyz.dtype), wh)
    pix = torch.einsum("Aij,ABj->ABi", intrs.to(world_xyz.dtype), cam)
    return pix[..., :2] / pix[..., -1:], cam[..., -1:]


@torch.no_grad()
def project_tracks(traj_e: torch.Tensor, intrs: torch.Tensor, extrs: torch.Tensor) -> torch.Tensor:
    """traj_e (1,T,N,3), intrs (1,V,T,3,3), extrs (1,V,T,3,4) -> 2-D tracks per view (1,V,T,N,2) (``traj2d_e`` of the evaluator)."""
    V = intrs.shape[1]
    out = [world_to_pixel(traj_e[0], intrs[0, v], extrs[0, v])[0] for v in range(V)]
    return torch.stack(out, 0)[None]
