"""Seeded synthetic clips and weights (numpy only, platform independent).

There is no network, dataset or checkpoint in this environment, so tests, ``bench.py`` and
the golden-vector generator all draw their inputs from here:

* ``make_clip``  - V pinhole cameras on a ring looking at a small analytic scene (ground plane
  + moving sphere), z-depth with +-1 cm uniform jitter so that no two kNN candidates are
  equidistant, RGB from a world-space texture, queries unprojected from random pixels.
* ``weight_for_key`` - the name-keyed weights recipe; ``fill_weights`` applies it to any module /
  dict whose keys follow the reference ``state_dict`` contract.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np


def make_cameras(V: int, T: int, H: int, W: int, radius: float = 3.0, height: float = 1.5):
    """intrs (V,T,3,3) and world->camera extrs (V,T,3,4), float64; static over T."""
    intr = np.array([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1.0]])
    intrs = np.broadcast_to(intr, (V, T, 3, 3)).copy()
    extrs = np.zeros((V, T, 3, 4))
    up = np.array([0.0, 0.0, 1.0])
    for v in range(V):
        th = 2 * math.pi * v / V + 0.3
        c = np.array([radius * math.cos(th), radius * math.sin(th), height])
        fwd = np.array([0.0, 0.0, 0.4]) - c
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        extrs[v, :, :, :3] = R
        extrs[v, :, :, 3] = -R @ c
    return intrs, extrs


def _render(intr, extr, t, H, W, rng, invalid_frac):
    R, tv = extr[:, :3], extr[:, 3]
    c = -R.T @ tv
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    dc = np.stack([(xs - intr[0, 2]) / intr[0, 0], (ys - intr[1, 2]) / intr[1, 1], np.ones_like(xs)], -1)
    dw = dc @ R  # R^T applied to each direction
    # ground plane z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        lam_p = np.where(dw[..., 2] < -1e-6, -c[2] / dw[..., 2], np.inf)
    # moving sphere
    sc = np.array([0.4 * math.sin(0.3 * t), 0.3 * math.cos(0.2 * t), 0.6])
    oc = c - sc
    a = (dw * dw).sum(-1)
    b = 2 * (dw * oc).sum(-1)
    cc = (oc * oc).sum() - 0.6 ** 2
    disc = b * b - 4 * a * cc
    lam_s = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    lam_s = np.where(lam_s > 0, lam_s, np.inf)
    lam = np.minimum(lam_p, lam_s)
    lam = np.where(np.isfinite(lam), lam, 6.0)
    hit = c + lam[..., None] * dw
    lam = lam * (1.0 + 0.02 * np.sin(3 * hit[..., 0]) * np.cos(2 * hit[..., 1]))
    # jitter AFTER the clip: a clipped far plane would otherwise be a perfect lattice full of
    # equidistant kNN candidates
    depth = np.clip(lam, 0.5, 5.9) + rng.uniform(-0.01, 0.01, size=lam.shape)
    if invalid_frac > 0:
        depth = np.where(rng.uniform(size=depth.shape) < invalid_frac, 0.0, depth)
    x, y, z = hit[..., 0], hit[..., 1], hit[..., 2]
    rgb = np.stack([127 + 100 * np.sin(5 * x + 1) * np.cos(4 * y),
                    127 + 100 * np.sin(3 * y + 8 * z),
                    127 + 100 * np.cos(6 * z + 2 * x)], 0)
    rgb = np.clip(np.rint(rgb + rng.normal(0, 8, size=rgb.shape)), 0, 255)
    return rgb, depth


def make_clip(seed: int, V: int, T: int, H: int, W: int, N: int, late_queries: bool = False,
              invalid_frac: float = 0.0, query_frames: Sequence[int] = (3, 7, 13), frame_period: Optional[int] = None,
              rgb_dtype=np.float32) -> Dict[str, np.ndarray]:
    """Seeded clip with the predictor's input layout (batch dim 1), all float32.

    rgbs (1,V,T,3,H,W) integer-valued in [0,255]; depths (1,V,T,1,H,W) metres; intrs (1,V,T,3,3);
    extrs (1,V,T,3,4); query_points (1,N,4) = (t, x, y, z) in world space.  With
    ``late_queries`` a quarter of the queries start at a frame from ``query_frames``.  ``frame_period`` renders only the
    first ``frame_period`` frames and repeats them (long high-resolution clips for throughput / property tests);
    ``rgb_dtype=np.uint8`` stores the (integer-valued) frames as bytes.
    """
    rng = np.random.default_rng(seed)
    intrs, extrs = make_cameras(V, T, H, W)
    rgbs = np.zeros((V, T, 3, H, W), rgb_dtype)
    depths = np.zeros((V, T, 1, H, W), np.float32)
    for v in range(V):
        for t in range(T):
            if frame_period is not None and t >= frame_period:
                rgbs[v, t] = rgbs[v, t % frame_period]
                depths[v, t] = depths[v, t % frame_period]
                continue
            rgb, d = _render(intrs[v, t], extrs[v, t], t, H, W, rng, invalid_frac)
            rgbs[v, t] = rgb
            depths[v, t, 0] = d
    q = np.zeros((N, 4), np.float64)
    frames = [f for f in query_frames if f < T] or [0]
    for n in range(N):
        tq = 0
        if late_queries and rng.uniform() < 0.25:
            tq = int(frames[rng.integers(len(frames))])
        for _ in range(64):
            v = int(rng.integers(V))
            px = int(rng.integers(W // 16, W - W // 16))
            py = int(rng.integers(H // 16, H - H // 16))
            d = float(depths[v, tq, 0, py, px])
            if d > 0:
                break
        K, E = intrs[v, tq], extrs[v, tq]
        cam = np.linalg.inv(K) @ np.array([px, py, 1.0]) * d
        world = E[:, :3].T @ (cam - E[:, 3])
        q[n] = [tq, *world]
    return {"rgbs": rgbs[None], "depths": depths[None], "intrs": intrs[None].astype(np.float32),
            "extrs": extrs[None].astype(np.float32), "query_points": q[None].astype(np.float32)}


def weight_for_key(key: str, shape: Tuple[int, ...], seed: int = 0, delta_scale: float = 0.005) -> np.ndarray:
    """One tensor of the seeded weights recipe, float32."""
    rng = np.random.default_rng((zlib.crc32(key.encode()) ^ seed) & 0xFFFFFFFF)
    if key.endswith("virual_tracks"):
        a = rng.standard_normal(shape)
    elif key.endswith(".bias"):
        a = 0.02 * rng.standard_normal(shape)
    elif "norm" in key.split(".")[-2] and key.endswith(".weight"):
        a = 1.0 + 0.02 * rng.standard_normal(shape)
    else:
        a = rng.standard_normal(shape) / math.sqrt(int(np.prod(shape[1:])))
        if key == "updateformer.flow_head.4.weight":
            a = a * delta_scale
    if key == "updateformer.flow_head.4.bias":
        a = a * delta_scale
    return np.ascontiguousarray(a, dtype=np.float32)


def make_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, delta_scale: float = 0.005):
    """key -> float32 ndarray for every entry of ``shapes``."""
    return {k: weight_for_key(k, tuple(s), seed, delta_scale) for k, s in sorted(shapes.items())}
