"""CPU restatement (plain PyTorch fp32) of the reference's multi-view tracking forward path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity status: PINNED.  The functions
below are checked in this container against the imported reference (tests/golden/make_golden.py
generates tests/golden/*.npz from /root/reference; tests/test_oracle_golden.py re-checks the
oracle against those files on every run).  The reference has no tests / golden vectors of its
own for this path.  Round 2 adds reference runs under
``torch.autocast("cpu", dtype=torch.bfloat16)`` (tests/golden/make_golden_r2.py, the arithmetic of
BASELINE config C3): run under the same autocast context this file is BIT-IDENTICAL to them
(``knn_mode="cdist"``; 0.0 max difference on tracks and visibilities), including the reference's
autocast quirks (bf16 point clouds, bf16 ``ffeats``).

Every function cites the reference file:line it follows (paths relative to /root/reference).
The code is a functional restatement over a flat ``weights`` dict that uses the reference's
``state_dict`` keys; it shares no code with the reference.

kNN semantics.  The reference has two backends (mvtracker.py:26-90): ``pointops.knn_query``
(exact fp32 squared distances, used on GPU) and ``torch.cdist + topk`` (CPU fallback; cdist
uses the matmul expansion for large inputs).  ``knn_mode="exact"`` restates the former with a
fully specified arithmetic: d2 = fma(dz,dz, fma(dy,dy, dx*dx)) in fp32, ascending by
(d2, index).  ``knn_mode="cdist"`` calls the very same torch ops
as the reference's fallback, so that the oracle can be pinned bit-for-bit against the reference
as imported here.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration + weights recipe
# --------------------------------------------------------------------------------------
@dataclass
class TrackerConfig:
    """Constructor arguments of the reference model (mvtracker.py:94-113)."""
    sliding_window_len: int = 12
    stride: int = 4
    fmaps_dim: int = 128
    add_space_attn: bool = True
    num_heads: int = 6
    hidden_size: int = 256  # configs/model/mvtracker.yaml:13 (class default is 384)
    space_depth: int = 6
    time_depth: int = 6
    num_virtual_tracks: int = 64
    corr_n_groups: int = 1
    corr_n_levels: int = 4
    corr_neighbors: int = 16
    corr_add_neighbor_offset: bool = True
    corr_add_neighbor_xyz: bool = False
    dim_head: int = 48  # cotracker2/blocks.py:247
    mlp_ratio: float = 4.0
    flow_embed_dim: int = 64  # mvtracker.py:120

    @property
    def corr_feat_per_neighbor(self) -> int:
        return self.corr_n_groups + 3 * int(self.corr_add_neighbor_offset) + 3 * int(self.corr_add_neighbor_xyz)

    @property
    def token_dim(self) -> int:  # mvtracker.py:130-149
        return ((self.flow_embed_dim + 1) * 3
                + self.corr_neighbors * self.corr_n_levels * self.corr_feat_per_neighbor
                + self.fmaps_dim + 2)

    @property
    def out_dim(self) -> int:
        return 3 + self.fmaps_dim


def state_dict_shapes(cfg: TrackerConfig) -> Dict[str, Tuple[int, ...]]:
    """Shapes of the 323 tensors of the reference ``state_dict``."""
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k):
        s[name + ".weight"] = (cout, cin, k, k)
        s[name + ".bias"] = (cout,)

    def lin(name, cout, cin):
        s[name + ".weight"] = (cout, cin)
        s[name + ".bias"] = (cout,)

    # encoder: spatracker/blocks.py:130-211
    conv("fnet.conv1", 64, 3, 7)
    cin = 64
    for li, cout in zip((1, 2, 3, 4), (64, 96, 128, 128)):
        for bi in (0, 1):
            conv(f"fnet.layer{li}.{bi}.conv1", cout, cin if bi == 0 else cout, 3)
            conv(f"fnet.layer{li}.{bi}.conv2", cout, cout, 3)
            if bi == 0 and li != 1:
                conv(f"fnet.layer{li}.0.downsample.0", cout, cin, 1)
        cin = cout
    conv("fnet.conv2", cfg.fmaps_dim * 2, 128 + 128 + 96 + 64, 3)
    conv("fnet.conv3", cfg.fmaps_dim, cfg.fmaps_dim * 2, 1)

    # updater: cotracker2/blocks.py:340-434
    h = cfg.hidden_size
    inner = cfg.num_heads * cfg.dim_head
    mlp = int(h * cfg.mlp_ratio)
    u = "updateformer."
    s[u + "virual_tracks"] = (1, cfg.num_virtual_tracks, 1, h)
    lin(u + "input_transform", h, cfg.token_dim)
    lin(u + "flow_head.0", cfg.out_dim, h)
    lin(u + "flow_head.2", cfg.out_dim, cfg.out_dim)
    lin(u + "flow_head.4", cfg.out_dim, cfg.out_dim)

    def attn(prefix):
        lin(prefix + ".to_q", inner, h)
        lin(prefix + ".to_kv", 2 * inner, h)
        lin(prefix + ".to_out", h, inner)

    def mlp_(prefix):
        lin(prefix + ".fc1", mlp, h)
        lin(prefix + ".fc2", h, mlp)

    for i in range(cfg.time_depth):
        attn(f"{u}time_blocks.{i}.attn")
        mlp_(f"{u}time_blocks.{i}.mlp")
    if cfg.add_space_attn:
        for i in range(cfg.space_depth):
            attn(f"{u}space_virtual_blocks.{i}.attn")
            mlp_(f"{u}space_virtual_blocks.{i}.mlp")
            for kind in ("space_point2virtual_blocks", "space_virtual2point_blocks"):
                p = f"{u}{kind}.{i}"
                s[p + ".norm_context.weight"] = (h,)
                s[p + ".norm_context.bias"] = (h,)
                attn(p + ".cross_attn")
                mlp_(p + ".mlp")
    s["ffeats_norm.weight"] = (cfg.fmaps_dim,)
    s["ffeats_norm.bias"] = (cfg.fmaps_dim,)
    lin("ffeats_updater.0", cfg.fmaps_dim, cfg.fmaps_dim)
    lin("vis_predictor.0", 1, cfg.fmaps_dim)
    return s


def make_weights(cfg: TrackerConfig, seed: int = 0, delta_scale: float = 0.005) -> Dict[str, Tensor]:
    """Seeded synthetic weights (SURVEY.md section 8c recipe); identical on every platform.

    Each tensor is drawn from ``np.random.default_rng(crc32(key) ^ seed)``: matrices / conv
    kernels ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2), norm scales 1 + N(0, 0.02^2), virtual
    tracks ~ N(0,1).  The last ``flow_head`` layer is scaled by ``delta_scale`` so that one
    refinement step moves a track by centimetres (upstream init, cotracker2/blocks.py:443-451,
    would make deltas ~1e-5 and hide errors).
    """
    out: Dict[str, Tensor] = {}
    for key, shape in sorted(state_dict_shapes(cfg).items()):
        rng = np.random.default_rng((zlib.crc32(key.encode()) ^ seed) & 0xFFFFFFFF)
        if key.endswith("virual_tracks"):
            a = rng.standard_normal(shape)
        elif key.endswith(".bias"):
            a = 0.02 * rng.standard_normal(shape)
        elif "norm" in key.split(".")[-2] and key.endswith(".weight"):
            a = 1.0 + 0.02 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) / math.sqrt(fan_in)
            if key == "updateformer.flow_head.4.weight":
                a = a * delta_scale
        if key == "updateformer.flow_head.4.bias":
            a = a * delta_scale
        out[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


# --------------------------------------------------------------------------------------
# encoder  (spatracker/blocks.py:70-128 ResidualBlock, :130-284 BasicEncoder)
# --------------------------------------------------------------------------------------
def _conv(W, name, x, stride=1, pad=0):
    return F.conv2d(x, W[name + ".weight"], W[name + ".bias"], stride=stride, padding=pad)


def _inorm(x):  # nn.InstanceNorm2d defaults: eps 1e-5, no affine, no running stats
    return F.instance_norm(x, eps=1e-5)


def _res_block(W, p, x, stride):  # spatracker/blocks.py:119-128
    y = F.relu(_inorm(_conv(W, p + ".conv1", x, stride, 1)))
    y = F.relu(_inorm(_conv(W, p + ".conv2", y, 1, 1)))
    if (p + ".downsample.0.weight") in W:
        x = _inorm(_conv(W, p + ".downsample.0", x, stride, 0))
    return F.relu(x + y)


def encoder_stages(W, x: Tensor) -> List[Tensor]:
    """conv1/IN/ReLU and the four residual stages; x (n,3,H,W) -> [a,b,c,d] (blocks.py:214-253)."""
    x = F.relu(_inorm(_conv(W, "fnet.conv1", x, 2, 3)))
    outs = []
    for li, stride in zip((1, 2, 3, 4), (1, 2, 2, 2)):
        x = _res_block(W, f"fnet.layer{li}.0", x, stride)
        x = _res_block(W, f"fnet.layer{li}.1", x, 1)
        outs.append(x)
    return outs


def encoder(W, x: Tensor, stride: int = 4) -> Tensor:
    """BasicEncoder.forward, instance-norm variant (blocks.py:214-284): (n,3,H,W) -> (n,128,H/4,W/4)."""
    H, Wd = x.shape[-2:]
    size = (H // stride, Wd // stride)
    feats = [F.interpolate(t, size, mode="bilinear", align_corners=True) for t in encoder_stages(W, x)]
    y = _conv(W, "fnet.conv2", torch.cat(feats, 1), 1, 1)
    y = F.relu(_inorm(y))
    return _conv(W, "fnet.conv3", y, 1, 0)


# --------------------------------------------------------------------------------------
# geometry / pyramid  (model_utils.py:420-482)
# --------------------------------------------------------------------------------------
def invert_cameras(intrs: Tensor, extrs: Tensor) -> Tuple[Tensor, Tensor]:
    """K^-1 (...,3,3) and [R|t]^-1 (...,4,4) in fp32 (model_utils.py:453-457)."""
    kinv = torch.inverse(intrs.float()).type(intrs.dtype)
    sq = torch.eye(4).repeat(*extrs.shape[:-2], 1, 1)
    sq[..., :3, :] = extrs
    einv = torch.inverse(sq.float()).type(extrs.dtype)
    return kinv, einv


def pointcloud_level(fmaps: Tensor, depths: Tensor, intrs: Tensor, extrs: Tensor, stride: int = 4,
                     level: int = 0, return_valid: bool = False):
    """init_pointcloud_from_rgbd (model_utils.py:420-482).

    fmaps (B,V,S,C,H,W), depths (B,V,S,1,H,W) at the strided resolution -> xyz (B*S, V*h*w, 3),
    fvec (B*S, V*h*w, C) with h = H / 2^level.  Features are average-pooled (:440), depth is
    nearest-subsampled (:443-444), the pixel grid is (i+0.5)*stride*2^level-0.5 (:462-466).
    """
    B, V, S, C, H, W = fmaps.shape
    f = fmaps.reshape(B * V * S, C, H, W)
    d = depths.reshape(B * V * S, 1, H, W)
    for _ in range(level):
        f = F.avg_pool2d(f, 2, stride=2)
        d = F.interpolate(d, scale_factor=0.5, mode="nearest")
    h, w = H // 2 ** level, W // 2 ** level
    f = f.reshape(B, V, S, C, h, w)
    d = d.reshape(B, V, S, 1, h, w)
    st = stride * 2 ** level
    kinv, einv = invert_cameras(intrs, extrs)
    ys = (torch.arange(0, h) + 0.5) * st - 0.5
    xs = (torch.arange(0, w) + 0.5) * st - 0.5
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    pix = torch.stack([gx, gy, torch.ones_like(gx)], -1).to(f.dtype)
    cam = torch.einsum("BVSij,HWj->BVSHWi", kinv, pix) * d[..., 0, :, :, None]
    cam_h = torch.cat([cam, torch.ones_like(cam[..., :1])], -1)
    world_h = torch.einsum("BVSij,BVSHWj->BVSHWi", einv, cam_h)
    world = world_h[..., :3] / world_h[..., 3:]
    xyz = world.permute(0, 2, 1, 3, 4, 5).reshape(B * S, V * h * w, 3)
    fvec = f.permute(0, 2, 1, 4, 5, 3).reshape(B * S, V * h * w, C)
    if return_valid:
        return xyz, fvec, d.permute(0, 2, 1, 3, 4, 5).reshape(B * S, V * h * w) > 0
    return xyz, fvec


# --------------------------------------------------------------------------------------
# kNN + point-cloud correlation  (mvtracker.py:26-90, 800-846)
# --------------------------------------------------------------------------------------
def knn_exact(k: int, ref: Tensor, query: Tensor, chunk: int = 128) -> Tuple[Tensor, Tensor]:
    """Exact fp32 kNN (pointops semantics, mvtracker.py:55-72): ref (B,P,3), query (B,M,3).

    d2 = fma(dz, dz, fma(dy, dy, dx*dx)) in fp32 with dx = ref - query -- the arithmetic both
    reference backends perform (pointops' CUDA kernel under nvcc's default FMA contraction;
    torch's non-matmul cdist on CPU, verified bit-for-bit in this container: sqrt(d2) equals
    ``torch.cdist(compute_mode='donot_use_mm_for_euclid_dist')`` on every pair tried).  The
    fp32 FMA is emulated through float64 (product of two fp32 is exact in fp64; the residual
    double-rounding case has probability ~2^-29 per operation).  Neighbours ascending by
    (d2, index).  Returns (d2 (B,M,k) fp32, idx (B,M,k) int64).
    """
    B, P, _ = ref.shape
    M = query.shape[1]
    d2_out = torch.empty(B, M, k, dtype=torch.float32)
    idx_out = torch.empty(B, M, k, dtype=torch.int64)
    ar = torch.arange(P, dtype=torch.int64)
    for b in range(B):
        r = ref[b].float()
        for m0 in range(0, M, chunk):
            q = query[b, m0:m0 + chunk].float()
            dx = r[None, :, 0] - q[:, None, 0]
            dy = (r[None, :, 1] - q[:, None, 1]).double()
            dz = (r[None, :, 2] - q[:, None, 2]).double()
            acc = (dy * dy + (dx * dx).double()).float()
            d2 = (dz * dz + acc.double()).float()
            # non-negative fp32 bit patterns are monotone -> (bits << 32 | index) orders by (d2, index)
            key = (d2.view(torch.int32).to(torch.int64) << 32) | ar[None, :]
            key = torch.topk(key, k, dim=1, largest=False, sorted=True).values
            idx = key & 0xFFFFFFFF
            idx_out[b, m0:m0 + chunk] = idx
            d2_out[b, m0:m0 + chunk] = torch.gather(d2, 1, idx)
    return d2_out, idx_out


def knn_cdist(k: int, ref: Tensor, query: Tensor) -> Tuple[Tensor, Tensor]:
    """The reference's CPU fallback, same torch ops (mvtracker.py:75-79)."""
    d = torch.cdist(query, ref, p=2)
    return torch.topk(d, k, dim=-1, largest=False, sorted=True)


def knn(k, ref, query, mode="exact"):
    return knn_exact(k, ref, query) if mode == "exact" else knn_cdist(k, ref, query)


def corr_sample(xyz: Tensor, fvec: Tensor, targets: Tensor, coords: Tensor, k: int = 16, groups: int = 1,
                add_offset: bool = True, add_xyz: bool = False, knn_mode: str = "exact",
                return_idx: bool = False):
    """PointcloudCorrBlock.corr_sample, filter_invalid=False (mvtracker.py:810-846).

    xyz (B,P,3), fvec (B,P,C), targets (B,M,C), coords (B,M,3) -> (B,M,k,groups+3[+3]).
    """
    B, P, C = fvec.shape
    M = targets.shape[1]
    _, idx = knn(k, xyz, coords, knn_mode)
    bidx = torch.arange(B)[:, None, None]
    nxyz = xyz[bidx, idx]
    nf = fvec[bidx, idx]
    corr = torch.einsum("BMGc,BMKGc->BMKG", targets.view(B, M, groups, -1), nf.view(B, M, k, groups, -1))
    corr = corr / ((C / groups) ** 0.5)
    out = corr
    if add_offset:
        out = torch.cat([corr, nxyz - coords[..., None, :]], -1)
    if add_xyz:
        out = torch.cat([out, nxyz], -1)
    return (out, idx) if return_idx else out


# --------------------------------------------------------------------------------------
# secondary operator: bilinear-window correlation (spatracker/blocks.py:423-449, 492-533, 604-619)
# --------------------------------------------------------------------------------------
def window_corr_pyramid(fmaps: Tensor, num_levels: int = 4) -> List[Tensor]:
    """CorrBlock.__init__ (blocks.py:423-449): fmaps (B,S,C,H,W) -> avg-pooled pyramid."""
    B, S, C, H, W = fmaps.shape
    pyr = [fmaps]
    for _ in range(num_levels - 1):
        f = F.avg_pool2d(pyr[-1].reshape(B * S, C, *pyr[-1].shape[-2:]), 2, stride=2)
        pyr.append(f.reshape(B, S, C, *f.shape[-2:]))
    return pyr


def window_corr_sample(pyr: List[Tensor], targets: Tensor, coords: Tensor, radius: int = 4) -> Tensor:
    """CorrBlock.corr_sample without depth pyramid (blocks.py:492-533).

    targets (B,S,N,C), coords (B,S,N,2) in level-0 pixels -> (B,S,N,L*(2r+1)^2).  At each level
    the (2r+1)^2 grid centred on coords/2^l is sampled with grid_sample(align_corners=True,
    zero padding) and dotted with the target, / sqrt(C).  delta[...,0] (the dy linspace) is added
    to x and delta[...,1] to y, exactly as the reference does (:502-507).
    """
    B, S, N, C = targets.shape
    r = radius
    D = (2 * r + 1) ** 2
    outs = []
    for i, f in enumerate(pyr):
        H, W = f.shape[-2:]
        lin = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1)
        cl = coords.reshape(B * S * N, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        g = cl.view(B * S, 1, N * D, 2)
        gx = 2 * g[..., 0:1] / (W - 1) - 1
        gy = 2 * g[..., 1:2] / (H - 1) - 1
        smp = F.grid_sample(f.reshape(B * S, C, H, W), torch.cat([gx, gy], -1), align_corners=True)
        smp = smp.permute(0, 3, 1, 2).reshape(B * S * N, D, C)
        c = torch.matmul(targets.reshape(B * S * N, 1, C), smp.permute(0, 2, 1))
        c = c / torch.sqrt(torch.tensor(C).float())
        outs.append(c.view(B, S, N, D))
    return torch.cat(outs, -1).contiguous().float()


# --------------------------------------------------------------------------------------
# embeddings  (embeddings.py:35-50, 88-106, 134-161)
# --------------------------------------------------------------------------------------
def sincos_1d(dim: int, pos) -> np.ndarray:
    """get_1d_sincos_pos_embed_from_grid (embeddings.py:88-106): float64, [sin | cos]."""
    omega = np.arange(dim // 2, dtype=np.float64)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", np.asarray(pos).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def pos_embed_3d(dim_padded: int, xyz: Tensor) -> Tensor:
    """get_3d_sincos_pos_embed_from_grid (embeddings.py:35-50): xyz (M,3) fp32 -> (M, dim_padded) fp64."""
    a = xyz.detach().cpu().numpy()
    e = [sincos_1d(dim_padded // 3, a[:, i]) for i in range(3)]
    return torch.from_numpy(np.concatenate(e, axis=1))


def flow_embedding(flow: Tensor, C: int = 64) -> Tensor:
    """get_3d_embedding(cat_coords=True) (embeddings.py:134-161): (B,N,3) -> (B,N,3C+3)."""
    div = (torch.arange(0, C, 2, dtype=torch.float32) * (1000.0 / C)).reshape(1, 1, C // 2)
    parts = []
    for i in range(3):
        a = flow[:, :, i:i + 1] * div
        pe = torch.zeros(*flow.shape[:2], C, dtype=torch.float32)
        pe[:, :, 0::2] = torch.sin(a)
        pe[:, :, 1::2] = torch.cos(a)
        parts.append(pe)
    return torch.cat(parts + [flow], dim=2)


def window_embeddings(cfg: TrackerConfig, coords0: Tensor, S: int) -> Tuple[Tensor, Tensor]:
    """pos_embed (N,1,D) from frame-0 coords (N,3) and times_embed (1,S,D) (mvtracker.py:324-344)."""
    D = cfg.token_dim
    d3 = D + (-D) % 6
    pos = pos_embed_3d(d3, coords0).float()[:, :D].unsqueeze(1)
    d2 = D + (D % 2)
    times = torch.linspace(0, S - 1, S).reshape(S, 1) / S
    te = torch.from_numpy(sincos_1d(d2, times.numpy()))[None].float()[:, :, :D]
    return pos, te


# --------------------------------------------------------------------------------------
# updater transformer  (cotracker2/blocks.py:37-67, 246-337, 455-494)
# --------------------------------------------------------------------------------------
def _lin(W, name, x):
    return F.linear(x, W[name + ".weight"], W[name + ".bias"])


def _attention(W, p, x, ctx, heads):  # FlashAttention.forward, blocks.py:258-271
    B, N1, _ = x.shape
    N2 = ctx.shape[1]
    q = _lin(W, p + ".to_q", x).reshape(B, N1, heads, -1).transpose(1, 2)
    k, v = _lin(W, p + ".to_kv", ctx).chunk(2, dim=-1)
    k = k.reshape(B, N2, heads, -1).transpose(1, 2)
    v = v.reshape(B, N2, heads, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    return _lin(W, p + ".to_out", o.transpose(1, 2).reshape(B, N1, -1))


def _ln(x, w=None, b=None, eps=1e-6):
    return F.layer_norm(x, x.shape[-1:], w, b, eps)


def _mlp(W, p, x):  # Mlp with tanh-GELU, blocks.py:289-295
    return _lin(W, p + ".fc2", F.gelu(_lin(W, p + ".fc1", x), approximate="tanh"))


def attn_block(W, p, x, heads):  # AttnBlock.forward, blocks.py:297-300
    n = _ln(x)
    x = x + _attention(W, p + ".attn", n, n, heads)
    return x + _mlp(W, p + ".mlp", _ln(x))


def cross_block(W, p, x, ctx, heads):  # CrossAttnBlock.forward, blocks.py:334-337
    c = _ln(ctx, W[p + ".norm_context.weight"], W[p + ".norm_context.bias"], eps=1e-5)
    x = x + _attention(W, p + ".cross_attn", _ln(x), c, heads)
    return x + _mlp(W, p + ".mlp", _ln(x))


def update_former(W, x: Tensor, cfg: TrackerConfig) -> Tensor:
    """EfficientUpdateFormer.forward (blocks.py:455-494): x (B,N,T,token_dim) -> (B,N,T,3+C)."""
    u = "updateformer."
    tok = _lin(W, u + "input_transform", x)
    B, _, T, _ = tok.shape
    tok = torch.cat([tok, W[u + "virual_tracks"].repeat(B, 1, T, 1)], dim=1)
    N = tok.shape[1]
    nv = cfg.num_virtual_tracks
    every = cfg.time_depth // cfg.space_depth if cfg.add_space_attn else 0
    j = 0
    for i in range(cfg.time_depth):
        t = attn_block(W, f"{u}time_blocks.{i}", tok.contiguous().view(B * N, T, -1), cfg.num_heads)
        tok = t.view(B, N, T, -1)
        if cfg.add_space_attn and i % every == 0:
            sp = tok.permute(0, 2, 1, 3).contiguous().view(B * T, N, -1)
            pt, vt = sp[:, :N - nv], sp[:, N - nv:]
            vt = cross_block(W, f"{u}space_virtual2point_blocks.{j}", vt, pt, cfg.num_heads)
            vt = attn_block(W, f"{u}space_virtual_blocks.{j}", vt, cfg.num_heads)
            pt = cross_block(W, f"{u}space_point2virtual_blocks.{j}", pt, vt, cfg.num_heads)
            tok = torch.cat([pt, vt], dim=1).view(B, T, N, -1).permute(0, 2, 1, 3)
            j += 1
    tok = tok[:, :N - nv]
    y = F.relu(_lin(W, u + "flow_head.0", tok))
    y = F.relu(_lin(W, u + "flow_head.2", y))
    return _lin(W, u + "flow_head.4", y)


# --------------------------------------------------------------------------------------
# refinement loop for one window  (mvtracker.py:244-410)
# --------------------------------------------------------------------------------------
def assemble_tokens(cfg, coords: Tensor, fcorrs: Tensor, ffeats: Tensor, mask_vis: Tensor,
                    pos: Tensor, te: Tensor) -> Tensor:
    """Token assembly (mvtracker.py:374-387): coords (1,S,N,3), fcorrs (1,S,N,LKF), ffeats (1,S,N,C),
    mask_vis (N,S,2) -> x (1,N,S,token_dim)."""
    _, S, N, _ = coords.shape
    fc = fcorrs.permute(0, 2, 1, 3).reshape(N, S, -1)
    fl = flow_embedding((coords - coords[:, 0:1]).permute(0, 2, 1, 3).reshape(N, S, 3), cfg.flow_embed_dim)
    ff = ffeats.permute(0, 2, 1, 3).reshape(N, S, -1)
    x = torch.cat([fl, fc, ff, mask_vis], dim=2) + pos + te
    return x[None]


def apply_delta(W, cfg, delta: Tensor, coords: Tensor, ffeats: Tensor) -> Tuple[Tensor, Tensor]:
    """Track / feature update (mvtracker.py:392-399): delta (N,S,3+C)."""
    N, S, _ = delta.shape
    C = cfg.fmaps_dim
    d_coord = delta[:, :, :3].reshape(1, N, S, 3).permute(0, 2, 1, 3)
    df = F.group_norm(delta[:, :, 3:3 + C].reshape(-1, C), 1, W["ffeats_norm.weight"], W["ffeats_norm.bias"], 1e-5)
    df = F.gelu(_lin(W, "ffeats_updater.0", df)).view(1, N, S, C).permute(0, 2, 1, 3)
    return coords + d_coord, ffeats + df


def refine_window(W, cfg: TrackerConfig, fmaps: Tensor, depths: Tensor, intrs: Tensor, extrs: Tensor,
                  coords_init: Tensor, vis_init: Tensor, track_mask: Tensor, feat_init: Tensor,
                  iters: int = 4, knn_mode: str = "exact", trace: Optional[dict] = None):
    """MVTracker.forward_iteration (mvtracker.py:244-410).

    fmaps (1,V,S,C,H,W), depths (1,V,S,1,H,W), intrs (1,V,S,3,3), extrs (1,V,S,3,4),
    coords_init (1,S,N,3), vis_init (1,S,N,1), track_mask (1,S,N,1) bool, feat_init (1,S,N,C).
    Returns ([coords per iteration], vis logits (1,S,N)).
    """
    B, V, S, C, H, Wd = fmaps.shape
    assert B == 1
    N = coords_init.shape[2]
    assert bool(track_mask.any(1).all())
    clouds = [pointcloud_level(fmaps, depths, intrs, extrs, cfg.stride, lvl) for lvl in range(cfg.corr_n_levels)]
    coords = coords_init.clone()
    pos, te = window_embeddings(cfg, coords[0, 0], S)
    ffeats = feat_init.clone()
    mask_vis = torch.cat([track_mask, vis_init], dim=3).permute(0, 2, 1, 3).reshape(N, S, 2).float()
    preds = []
    for it in range(iters):
        fc = []
        for lvl, (xyz, fvec) in enumerate(clouds):
            o = corr_sample(xyz, fvec, ffeats.reshape(S, N, C), coords.reshape(S, N, 3), cfg.corr_neighbors,
                            cfg.corr_n_groups, cfg.corr_add_neighbor_offset, cfg.corr_add_neighbor_xyz, knn_mode,
                            return_idx=trace is not None)
            if trace is not None:
                o, idx = o
                trace.setdefault("knn_idx", []).append(idx)
            fc.append(o.reshape(1, S, N, -1))
        fcorrs = torch.cat(fc, dim=-1)
        x = assemble_tokens(cfg, coords, fcorrs, ffeats, mask_vis, pos, te)
        delta = update_former(W, x, cfg)[0]
        if trace is not None:
            trace.setdefault("fcorrs", []).append(fcorrs)
            trace.setdefault("tokens", []).append(x)
            trace.setdefault("delta", []).append(delta)
        coords, ffeats = apply_delta(W, cfg, delta, coords, ffeats)
        preds.append(coords.clone())
    vis = _lin(W, "vis_predictor.0", ffeats.reshape(S * N, C)).reshape(1, S, N)
    if trace is not None:
        trace["ffeats"] = ffeats
    return preds, vis


# --------------------------------------------------------------------------------------
# sliding-window tracker forward  (mvtracker.py:412-732)
# --------------------------------------------------------------------------------------
def tracker_forward(W, cfg: TrackerConfig, rgbs: Tensor, depths: Tensor, query_points: Tensor, intrs: Tensor,
                    extrs: Tensor, iters: int = 4, knn_mode: str = "exact", trace: Optional[dict] = None):
    """MVTracker.forward (mvtracker.py:412-732), inference branch.

    rgbs (1,V,T,3,H,W) in [0,255], depths (1,V,T,1,H,W), query_points (1,N,4)=(t,x,y,z),
    intrs (1,V,T,3,3), extrs (1,V,T,3,4).  Returns dict traj_e (1,T,N,3), vis_e (1,T,N) sigmoid,
    feat_init (1,S,N,C), plus vis_logits (1,T,N) and sort_inds for the integer-parity tests.
    """
    B, V, T, _, H, Wd = rgbs.shape
    assert B == 1
    N = query_points.shape[1]
    S, st, C = cfg.sliding_window_len, cfg.stride, cfg.fmaps_dim
    hs, ws = H // st, Wd // st
    qt = query_points[0, :, 0].long()  # :489 truncation toward zero
    qxyz = query_points[:, :, 1:]
    track_mask = (torch.arange(T)[None, :, None] >= qt[None, None, :]).unsqueeze(-1)  # :505-507
    sort_inds = torch.sort(qt, dim=0, descending=False, stable=True).indices  # :514 (order among equal t is free)
    inv_sort = torch.argsort(sort_inds, dim=0)
    qt_s = qt[sort_inds]
    qxyz_s = qxyz[:, sort_inds]
    coords_init = qxyz_s.unsqueeze(1).repeat(1, S, 1, 1).clone()  # :510
    vis_init = torch.full((1, S, N, 1), 10.0)  # :511
    track_mask = track_mask[:, :, sort_inds].clone()
    traj = torch.zeros(1, T, N, 3)
    vis_prob = torch.zeros(1, T, N)
    vis_logit = torch.zeros(1, T, N)
    w = int(qt_s.min())
    p0 = 0
    fmaps_seq = depths_seq = feat_init = None
    coords = vis = None
    windows = []
    while w < T - S // 2:  # :537
        p1 = int(torch.nonzero(qt_s < w + S)[-1]) + 1  # :538-540
        if fmaps_seq is None:
            t0 = w
        else:
            fmaps_seq, depths_seq, t0 = fmaps_seq[:, :, S // 2:], depths_seq[:, :, S // 2:], w + S // 2
        t1 = w + S
        d_new = F.interpolate(depths[:, :, t0:t1].reshape(-1, 1, H, Wd), scale_factor=1.0 / st, mode="nearest")
        d_new = d_new.reshape(1, V, -1, 1, hs, ws)  # :558-562
        f_new = encoder(W, (2 * (rgbs[:, :, t0:t1] / 255.0) - 1.0).reshape(-1, 3, H, Wd), st)  # :565-568
        f_new = F.interpolate(f_new, size=(hs, ws), mode="bilinear").reshape(1, V, -1, C, hs, ws)  # :569-573
        fmaps_seq = f_new if fmaps_seq is None else torch.cat([fmaps_seq, f_new], 2)
        depths_seq = d_new if depths_seq is None else torch.cat([depths_seq, d_new], 2)
        intrs_seq, extrs_seq = intrs[:, :, w:w + S], extrs[:, :, w:w + S]
        S_local = fmaps_seq.shape[2]
        if S_local < S:  # :598-604 repeat the last frame
            def pad(t):
                return torch.cat([t, t[:, :, -1:].repeat(1, 1, S - S_local, *([1] * (t.dim() - 3)))], 2)
            fmaps_seq, depths_seq, intrs_seq, extrs_seq = map(pad, (fmaps_seq, depths_seq, intrs_seq, extrs_seq))
        if p1 - p0 > 0:  # feature init by 1-NN in the frame's fused level-0 cloud, :607-645
            xyz, fvec = pointcloud_level(f_new, d_new, intrs[:, :, t0:t1], extrs[:, :, t0:t1], st, 0)
            f_init = torch.zeros(1, p1 - p0, C, dtype=f_new.dtype)  # :624-625 (bf16 under autocast)
            for t in range(t0, min(t1, T)):
                m = qt_s[p0:p1] == t
                if int(m.sum()) == 0:
                    continue
                _, nn_idx = knn(1, xyz[t - t0][None], qxyz_s[0, p0:p1][m][None], knn_mode)
                f_init[0, m] = fvec[t - t0][nn_idx[0, :, 0]]
                if trace is not None:
                    trace.setdefault("init_idx", []).append((t, nn_idx[0, :, 0].clone()))
            f_init = f_init[:, None].repeat(1, S, 1, 1)
            feat_init = f_init if feat_init is None else torch.cat([feat_init, f_init], 2)
        if p0 > 0:  # carry-over, :648-655 (vis is the previous window's LOGIT)
            last_c = coords[-1][:, S // 2:].clone()
            coords_init[:, :S // 2, :p0] = last_c
            coords_init[:, S // 2:, :p0] = last_c[:, -1:].repeat(1, S // 2, 1, 1)
            last_v = vis[:, S // 2:][..., None]
            vis_init[:, :S // 2, :p0] = last_v
            vis_init[:, S // 2:, :p0] = last_v[:, -1:].repeat(1, S // 2, 1, 1)
        tm = track_mask[:, w:w + S, :p1]
        if S_local < S:
            tm = torch.cat([tm, tm[:, -1:].repeat(1, S - S_local, 1, 1)], 1)
        wtrace = {} if trace is not None else None
        coords, vis = refine_window(W, cfg, fmaps_seq, depths_seq, intrs_seq, extrs_seq, coords_init[:, :, :p1],
                                    vis_init[:, :, :p1], tm, feat_init[:, :, :p1], iters, knn_mode, wtrace)
        traj[:, w:w + S, :p1] = coords[-1][:, :S_local]  # :692-693
        vis_prob[:, w:w + S, :p1] = torch.sigmoid(vis[:, :S_local])
        vis_logit[:, w:w + S, :p1] = vis[:, :S_local]
        track_mask[:, :w + S, :p1] = False  # :695
        windows.append((w, p1))
        if trace is not None:
            wtrace["fmaps_seq"] = fmaps_seq
            trace.setdefault("windows", []).append(wtrace)
        w += S // 2
        p0 = p1
    return {"traj_e": traj[:, :, inv_sort], "vis_e": vis_prob[:, :, inv_sort], "feat_init": feat_init,
            "vis_logits": vis_logit[:, :, inv_sort], "sort_inds": sort_inds, "inv_sort_inds": inv_sort,
            "windows": windows}


# --------------------------------------------------------------------------------------
# predictor wrapper  (evaluation_predictor_3dpt.py:47-414) and helpers (model_utils.py:81-165, 344-417)
# --------------------------------------------------------------------------------------
def bilinear_sample2d(im: Tensor, x: Tensor, y: Tensor) -> Tensor:
    """model_utils.py:81-165 for a 4-D image: im (B,C,H,W), x,y (B,N) pixels -> (B,C,N).

    Four clamped taps; the weights use the UNclamped integer corners."""
    B, C, H, W = im.shape
    x, y = x.float(), y.float()
    x0, y0 = torch.floor(x).int(), torch.floor(y).int()
    x1, y1 = x0 + 1, y0 + 1
    cx0, cx1 = x0.clamp(0, W - 1).long(), x1.clamp(0, W - 1).long()
    cy0, cy1 = y0.clamp(0, H - 1).long(), y1.clamp(0, H - 1).long()
    flat = im.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = torch.arange(B)[:, None]
    g = lambda yy, xx: flat[b, yy * W + xx]
    w00 = ((x1.float() - x) * (y1.float() - y)).unsqueeze(2)
    w01 = ((x - x0.float()) * (y1.float() - y)).unsqueeze(2)
    w10 = ((x1.float() - x) * (y - y0.float())).unsqueeze(2)
    w11 = ((x - x0.float()) * (y - y0.float())).unsqueeze(2)
    out = w00 * g(cy0, cx0) + w01 * g(cy0, cx1) + w10 * g(cy1, cx0) + w11 * g(cy1, cx1)
    return out.permute(0, 2, 1)


def grid_points(size: int, extent: Tuple[float, float], center: Optional[Tuple[float, float]] = None) -> Tensor:
    """get_points_on_a_grid (model_utils.py:361-417): (1, size*size, 2) as (x, y), margin W/64."""
    if size == 1:
        return torch.tensor([extent[1] / 2, extent[0] / 2])[None, None]
    if center is None:
        center = [extent[0] / 2, extent[1] / 2]
    m = extent[1] / 64
    ry = (m - extent[0] / 2 + center[0], extent[0] / 2 + center[0] - m)
    rx = (m - extent[1] / 2 + center[1], extent[1] / 2 + center[1] - m)
    gy, gx = torch.meshgrid(torch.linspace(*ry, size), torch.linspace(*rx, size), indexing="ij")
    return torch.stack([gx, gy], dim=-1).reshape(1, -1, 2)


def project_to_view(world_xyz: Tensor, intrs: Tensor, extrs: Tensor) -> Tuple[Tensor, Tensor]:
    """world_space_to_pixel_xy_and_camera_z (model_utils.py:344-358): (T,N,3),(T,3,3),(T,3,4)."""
    wh = torch.cat([world_xyz, torch.ones_like(world_xyz[..., :1])], -1)
    cam = torch.einsum("Aij,ABj->ABi", extrs, wh)
    pix = torch.einsum("Aij,ABj->ABi", intrs, cam)
    return pix[..., :2] / pix[..., -1:], cam[..., -1:]


def _unproject(pix_xy: Tensor, z: Tensor, kinv: Tensor, einv: Tensor) -> Tensor:
    ph = torch.cat([pix_xy, torch.ones_like(pix_xy[..., :1])], -1)
    cam = torch.einsum("Bij,BNj->BNi", kinv, ph) * z
    wh = torch.einsum("Bij,BNj->BNi", einv, torch.cat([cam, torch.ones_like(cam[..., :1])], -1))
    return wh[..., :3] / wh[..., 3:]


def predictor_prepare(rgbs, depths, query_points_3d, intrs, extrs, interp_shape=(384, 512), grid_size=5,
                      n_grids_per_view=1, uniform_pts=None):
    """Resize + support-grid synthesis of EvaluationPredictor.forward (evaluation_predictor_3dpt.py:59-120).

    Returns (rgbs, depths, intrs, support_points (1,M,4))."""
    B, V, T, _, Hr, Wr = rgbs.shape
    if B != 1:
        raise NotImplementedError
    if interp_shape is None:
        H, Wd = Hr, Wr
    else:
        H, Wd = interp_shape
        rgbs = F.interpolate(rgbs.reshape(-1, 3, Hr, Wr), (H, Wd), mode="nearest").reshape(B, V, T, 3, H, Wd)
        depths = F.interpolate(depths.reshape(-1, 1, Hr, Wr), (H, Wd), mode="nearest").reshape(B, V, T, 1, H, Wd)
        rs = torch.tensor([[Wd / Wr, 0, 0], [0, H / Hr, 0], [0, 0, 1]], dtype=intrs.dtype)
        intrs = torch.einsum("ij,BVTjk->BVTik", rs, intrs)
    kinv, einv = invert_cameras(intrs, extrs)
    support = torch.zeros((B, 0, 4))
    if grid_size > 0:
        pix = grid_points(grid_size, (H, Wd))
        pts = []
        for t in range(0, T, max(1, T // n_grids_per_view)):
            for v in range(V):
                z = bilinear_sample2d(depths[0, v, t][None], pix[..., 0], pix[..., 1]).permute(0, 2, 1)
                world = _unproject(pix, z, kinv[:, v, t], einv[:, v, t])
                pts.append(torch.cat([torch.ones_like(world[:, :, :1]) * t, world], dim=2))
        support = torch.cat([support, torch.cat(pts, dim=1)], dim=1)
    if uniform_pts is not None:
        # uniformly sampled support points (evaluation_predictor_3dpt.py:147-190); ``uniform_pts`` (n,3) = the rows of
        # get_uniformly_sampled_pts (:417-429).  Column 1 is what the reference calls y, column 2 what it calls x (:156-158).
        rows = []
        for i in range(uniform_pts.shape[0]):
            t = int(uniform_pts[i, 0].long())
            x, y = uniform_pts[i, 2].float().reshape(1, 1), uniform_pts[i, 1].float().reshape(1, 1)
            for v in range(V):
                z = bilinear_sample2d(depths[0, v, t][None], x, y).permute(0, 2, 1)
                world = _unproject(torch.stack([x, y], -1), z, kinv[:, v, t], einv[:, v, t])
                rows.append(torch.cat([torch.ones_like(world[:, :, :1]) * t, world], dim=2))
        if rows:
            support = torch.cat([support, torch.cat(rows, dim=1)], dim=1)
    return rgbs, depths, intrs, support


def predictor_forward(W, cfg, rgbs, depths, query_points_3d, intrs, extrs, interp_shape=(384, 512),
                      visibility_threshold=0.5, grid_size=5, n_grids_per_view=1, n_iters=6, knn_mode="exact", uniform_pts=None):
    """EvaluationPredictor.forward, joint mode (evaluation_predictor_3dpt.py:341-360, 410-414)."""
    n = query_points_3d.shape[1]
    rgbs, depths, intrs, support = predictor_prepare(rgbs, depths, query_points_3d, intrs, extrs, interp_shape,
                                                     grid_size, n_grids_per_view, uniform_pts)
    q = torch.cat([query_points_3d, support], dim=1)
    res = tracker_forward(W, cfg, rgbs, depths, q, intrs, extrs, iters=n_iters, knn_mode=knn_mode)
    vis = res["vis_e"][:, :, :n]
    return {"traj_e": res["traj_e"][:, :, :n], "vis_e": vis > visibility_threshold, "vis_e_as_prob": vis,
            "vis_logits": res["vis_logits"][:, :, :n], "support_points": support, "intrs": intrs}


def predictor_single_point_queries(depths, query_points_3d, intrs, extrs, support, local_grid_size=8, local_extent=50):
    """Per-query model inputs of single_point mode (evaluation_predictor_3dpt.py:191-256): for query i the rows are
    [query i ; local grid of every view around its projection, sampled at the query's own frame ; global support].
    depths (1,V,T,1,H,W) etc. are the predictor's (already resized) tensors.  Returns a list of (1,M_i,4)."""
    _, V, T, _, H, Wd = depths.shape
    n = query_points_3d.shape[1]
    kinv, einv = invert_cameras(intrs, extrs)
    qt = query_points_3d[..., :1].long()
    xyz = query_points_3d[..., 1:]
    wh = torch.cat([xyz, torch.ones_like(xyz[..., :1])], -1)
    cam = torch.einsum("BVTij,BNj->BVTNi", extrs, wh)  # :193-197
    ph = torch.einsum("BVTij,BVTNj->BVTNi", intrs, cam)
    pix = ph[..., :2] / ph[..., 2:]
    pix = pix[torch.arange(1)[:, None, None], torch.arange(V)[None, :, None], qt[:, None, :, 0], torch.arange(n)[None, None, :]]
    out = []
    for i in range(n):
        rows = [query_points_3d[:, i:i + 1]]
        if local_grid_size > 0:
            t = int(qt[0, i, 0])
            for v in range(V):
                g = grid_points(local_grid_size, (local_extent, local_extent), center=(pix[0, v, i, 1].item(), pix[0, v, i, 0].item()))
                ok = (g[0, :, 0] >= 0) & (g[0, :, 0] < Wd) & (g[0, :, 1] >= 0) & (g[0, :, 1] < H)  # :228-233
                if not bool(ok.any()):
                    continue
                g = g[:, ok]
                z = bilinear_sample2d(depths[0, v, t][None], g[..., 0], g[..., 1]).permute(0, 2, 1)
                world = _unproject(g, z, kinv[:, v, t], einv[:, v, t])
                rows.append(torch.cat([torch.ones_like(world[:, :, :1]) * t, world], dim=2))
        rows.append(support)
        out.append(torch.cat(rows, dim=1))
    return out


def predictor_forward_single_point(W, cfg, rgbs, depths, query_points_3d, intrs, extrs, interp_shape=(384, 512),
                                   visibility_threshold=0.5, grid_size=5, n_grids_per_view=1, local_grid_size=8, local_extent=50,
                                   n_iters=6, knn_mode="exact"):
    """EvaluationPredictor.forward, single_point mode (evaluation_predictor_3dpt.py:191-277, 410-414): one independent
    forward per query; only track 0 of each forward is kept."""
    n = query_points_3d.shape[1]
    T = rgbs.shape[2]
    rgbs, depths, intrs, support = predictor_prepare(rgbs, depths, query_points_3d, intrs, extrs, interp_shape, grid_size,
                                                     n_grids_per_view)
    qs = predictor_single_point_queries(depths, query_points_3d, intrs, extrs, support, local_grid_size, local_extent)
    traj = torch.zeros(1, T, n, 3)
    vis = torch.zeros(1, T, n)
    for i, q in enumerate(qs):
        res = tracker_forward(W, cfg, rgbs, depths, q, intrs, extrs, iters=n_iters, knn_mode=knn_mode)
        traj[:, :, i] = res["traj_e"][:, :, 0]
        vis[:, :, i] = res["vis_e"][:, :, 0]
    return {"traj_e": traj, "vis_e": vis > visibility_threshold, "vis_e_as_prob": vis, "per_query_inputs": qs}


def adapter_best_view(depths: Tensor, query_points: Tensor, intrs: Tensor, extrs: Tensor) -> Tensor:
    """View assignment of MonocularToMultiViewAdapter.forward (monocular_baselines.py:630-680).

    depths (V,T,1,H,W), query_points (N,4), intrs (V,T,3,3), extrs (V,T,3,4) -> (N,) int64."""
    V, T, _, H, W = depths.shape
    N = query_points.shape[0]
    qt = query_points[:, 0].long()
    xyz = query_points[:, 1:]
    xy = torch.zeros(V, N, 2)
    zc = torch.zeros(V, N, 1)
    dv = torch.zeros(V, N, 1)
    for t in qt.unique():
        m = qt == t
        wh = torch.cat([xyz[m], torch.ones_like(xyz[m][:, :1])], -1)
        cam = torch.einsum("Aij,Bj->ABi", extrs[:, t], wh)
        ph = torch.einsum("Aij,ABj->ABi", intrs[:, t], cam)
        xy[:, m] = ph[..., :2] / ph[..., 2:]
        zc[:, m] = cam[..., -1:]
        for v in range(V):
            dv[v, m] = bilinear_sample2d(depths[v, t][None], xy[v, m, 0][None], xy[v, m, 1][None])[0].permute(1, 0)
    outside = (xy[..., 0] < 0) | (xy[..., 0] >= W) | (xy[..., 1] < 0) | (xy[..., 1] >= H) | (zc[..., 0] < 0)
    dv = dv.clone()
    dv[outside] = -1e4
    return (dv - zc).argmax(0).squeeze(-1)
