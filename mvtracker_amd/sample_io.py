"""Sample / result files of the demo (reference demo.py:650, 922-929, 1086-1121).

Input NPZ: ``rgbs (V,T,3,H,W)``, ``depths (V,T,1,H,W)``, ``intrs (V,T,3,3)``, ``extrs (V,T,3,4 world->camera)``,
optional ``query_points (N,4 = t,x,y,z)``, ``camera_ids``, ``timestamps``, ``per_camera_timestamps``.
Result NPZ: ``tracks_3d (T,N,3)``, ``visibilities (T,N)``, ``query_points (N,4)`` plus the camera data.

The reference turns the whole clip into float32 host tensors first (4 bytes per colour value).  Here uint8 frames
stay uint8 from the file to the encoder's first kernel (``mvt_rgb_u8_to_nhwc4``): a quarter of the host memory, of the
PCIe traffic and of the HBM footprint, with bit-identical results ((float)u8 is exact).  Files are read with
``numpy.load(allow_pickle=False)`` -- nothing in them is executed.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

_OPTIONAL = ("camera_ids", "timestamps", "per_camera_timestamps")


def load_sample(path: str, device="cuda", temporal_stride: int = 1, spatial_downsample: int = 1) -> Dict[str, object]:
    """Read a sample NPZ and put it on ``device`` with a leading batch dimension, ready for
    ``EvaluationPredictor.forward`` / ``MVTracker.forward``.  ``temporal_stride`` / ``spatial_downsample`` follow
    demo.py:905-944 (frames ``[::stride]``, pixels ``[::ds]``, intrinsics rows 0-1 divided by ds)."""
    out: Dict[str, object] = {}
    with np.load(path, mmap_mode="r", allow_pickle=False) as z:
        keys = set(z.files)
        for k in ("rgbs", "depths", "intrs", "extrs"):
            if k not in keys:
                raise KeyError(f"{path}: missing '{k}'")
        ds = int(spatial_downsample)
        rgbs = np.asarray(z["rgbs"][:, ::temporal_stride, :, ::ds, ::ds])
        depths = np.asarray(z["depths"][:, ::temporal_stride, :, ::ds, ::ds], dtype=np.float32)
        intrs = np.array(z["intrs"][:, ::temporal_stride], dtype=np.float32)
        extrs = np.asarray(z["extrs"][:, ::temporal_stride], dtype=np.float32)
        if ds > 1:
            intrs[:, :, :2, :] /= ds
        qp = np.asarray(z["query_points"], dtype=np.float32) if "query_points" in keys else np.zeros((0, 4), np.float32)
        for k in _OPTIONAL:
            if k in keys and z[k].dtype.kind not in "O":
                out[k] = np.asarray(z[k])
    if rgbs.ndim != 5 or rgbs.shape[2] != 3 or depths.shape[:2] != rgbs.shape[:2] or depths.shape[2] != 1:
        raise ValueError(f"{path}: unexpected shapes rgbs {rgbs.shape}, depths {depths.shape}")
    if rgbs.dtype != np.uint8:  # float frames in [0, 255]: the model takes them as they are
        rgbs = rgbs.astype(np.float32)

    def put(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        if torch.device(device).type == "cuda":
            t = t.pin_memory().to(device, non_blocking=True)
        return t[None]

    out.update(rgbs=put(rgbs), depths=put(depths), intrs=put(intrs), extrs=put(extrs), query_points_3d=put(qp))
    return out


def save_result(path: str, traj_e: torch.Tensor, vis_e: torch.Tensor, sample: Dict[str, object], tracker: str = "mvtracker",
                temporal_stride: int = 1, spatial_downsample: int = 1, include_clip: bool = True) -> None:
    """Write the result NPZ of demo.py:1086-1121 (``tracks_3d`` (T,N,3), ``visibilities`` (T,N), ``query_points``, camera data)."""
    data = {
        "tracks_3d": traj_e.reshape(traj_e.shape[-3:]).float().cpu().numpy(),
        "visibilities": vis_e.reshape(vis_e.shape[-2:]).cpu().numpy(),
        "query_points": sample["query_points_3d"][0].float().cpu().numpy(),
        "tracker": np.asarray(tracker),
        "temporal_stride": np.asarray(temporal_stride),
        "spatial_downsample": np.asarray(spatial_downsample),
    }
    if include_clip:
        for k in ("rgbs", "depths", "intrs", "extrs"):
            data[k] = sample[k][0].cpu().numpy()
    for k in _OPTIONAL:
        if k in sample:
            data[k] = sample[k]
    np.savez_compressed(path, **data)
