"""Local model factory: the offline replacement for
``torch.hub.load("ethz-vlg/mvtracker", "mvtracker", pretrained=True, device=...)`` (reference demo.py:597-602).
"""
from __future__ import annotations

from typing import Optional

import torch

from .predictor import EvaluationPredictor
from .tracker import MVTracker

# constructor kwargs of configs/model/mvtracker.yaml:7-23
DEFAULT_MODEL_KWARGS = dict(
    sliding_window_len=12, stride=4, normalize_scene_in_fwd_pass=False, fmaps_dim=128, add_space_attn=True, num_heads=6,
    hidden_size=256, space_depth=6, time_depth=6, num_virtual_tracks=64, use_flash_attention=True, corr_n_groups=1,
    corr_n_levels=4, corr_neighbors=16, corr_add_neighbor_offset=True, corr_add_neighbor_xyz=False,
    corr_filter_invalid_depth=False)


def load_mvtracker(checkpoint: Optional[str] = None, device="cuda", pretrained: bool = False, **predictor_kwargs):
    """Build the predictor-wrapped tracker.  ``checkpoint`` is a local path to a reference checkpoint
    (either a bare state_dict or a Fabric dict with a "model" entry); it is read with
    ``torch.load(weights_only=True)``.  ``pretrained=True`` without a checkpoint is an error: nothing is downloaded."""
    model = MVTracker(**DEFAULT_MODEL_KWARGS)
    if checkpoint is not None:
        sd = torch.load(checkpoint, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "model" in sd and not any(k.startswith("fnet.") for k in sd):
            sd = sd["model"]
        model.load_state_dict(sd, strict=True)
    elif pretrained:
        raise FileNotFoundError("pretrained=True needs checkpoint=<local path>; this build never downloads weights")
    kw = dict(interp_shape=(384, 512), visibility_threshold=0.5, grid_size=5, n_iters=4)
    kw.update(predictor_kwargs)
    return EvaluationPredictor(model.to(device), **kw).to(device)
