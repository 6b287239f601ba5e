"""MVTracker on MI355X: host-side mirror of the reference model's interface.

Same constructor kwargs, ``forward`` signature, result dict and ``state_dict`` keys as
``mvtracker.models.core.mvtracker.mvtracker.MVTracker`` (reference mvtracker.py:93-181, 412-732),
so a Hydra ``_target_`` or a checkpoint can be pointed at this class unchanged.  All arithmetic
runs in libmvtracker_hip.so (see include/mvtracker_hip.h); this file only sequences kernel
launches, owns device buffers and does the per-window bookkeeping on tiny tensors.

Differences from the reference, all result-preserving:
  * every frame is encoded exactly once up front into a frame-major, channels-last "frame store"
    (features + world-space points for the 4 pyramid levels); windows are slices of it, the
    reference's repeat-last-frame padding is a clamped frame index inside the kernels;
  * track state is track-major ([N][S][.]) so the token / delta layouts need no permutes;
  * one host sync per call (the query frame indices), none inside the refinement loop; the NaN
    guard (reference mvtracker.py:401-404) is a device flag read once at the end.
"""
from __future__ import annotations

import contextlib
import logging
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import hip

log = logging.getLogger(__name__)


_param_epoch = [0]  # bumped whenever a Parameter OBJECT is (re)registered anywhere in a tracker's module tree


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state_dict keys."""

    def register_parameter(self, name, param):
        # (assigning a new nn.Parameter, load_state_dict(assign=True), ... all end here: the cached parameter list of
        #  MVTracker._signature must be rebuilt, or a stale packed weight set would be used silently)
        _param_epoch[0] += 1
        super().register_parameter(name, param)


def _insert(root: nn.Module, dotted: str, tensor: torch.Tensor) -> None:
    *path, leaf = dotted.split(".")
    m = root
    for p in path:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    m.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class _Shifted:
    """View of a tensor whose row ``i`` is ``base[i - shift]`` (slice access only)."""

    def __init__(self, base, shift):
        self.base, self.shift = base, shift

    def __getitem__(self, sl):
        return self.base[sl.start - self.shift:sl.stop - self.shift]


class _ClipImages:
    """Images img0 .. (frame-major numbering t * V + v) of the planar clip rgbs (V,T,3,H,W), fp32 or uint8 in [0, 255], by reference."""

    def __init__(self, rgbs, V, T, img0):
        self.rgbs, self.V, self.T, self.img0 = rgbs, V, T, img0


class MVTracker(nn.Module):
    def __init__(
            self,
            sliding_window_len=12,
            stride=4,
            normalize_scene_in_fwd_pass=False,
            fmaps_dim=128,
            add_space_attn=True,
            num_heads=6,
            hidden_size=384,
            space_depth=6,
            time_depth=6,
            num_virtual_tracks=64,
            use_flash_attention=True,
            corr_n_groups=1,
            corr_n_levels=4,
            corr_neighbors=16,
            corr_add_neighbor_offset=True,
            corr_add_neighbor_xyz=False,
            corr_filter_invalid_depth=False,
    ):
        super().__init__()
        if normalize_scene_in_fwd_pass:
            raise NotImplementedError("normalize_scene_in_fwd_pass is broken upstream (mvtracker.py:463-477)")
        if corr_filter_invalid_depth:
            raise NotImplementedError("corr_filter_invalid_depth=True gathers with mismatched indices upstream "
                                      "(mvtracker.py:820-829); only the default False is implemented")
        lanes = fmaps_dim // 8  # lanes of a bf16 feature row: the grouped dots stop their reduction at whole lanes
        if corr_n_groups < 1 or corr_n_groups & (corr_n_groups - 1) or corr_n_groups > lanes:
            raise NotImplementedError(f"corr_n_groups must be a power of two <= fmaps_dim / 8 = {lanes}")
        if not add_space_attn or time_depth != space_depth:
            raise NotImplementedError("only add_space_attn=True with time_depth == space_depth is implemented")
        if fmaps_dim not in (32, 64, 128, 256) or not (1 <= corr_neighbors <= 16) or stride != 4:
            raise NotImplementedError("fmaps_dim in {32,64,128,256}, corr_neighbors <= 16, stride 4")
        self.S = sliding_window_len
        self.stride = stride
        self.latent_dim = fmaps_dim
        self.flow_embed_dim = 64
        self.corr_n_levels = corr_n_levels
        self.corr_neighbors = corr_neighbors
        # correlation features per neighbour (mvtracker.py:136-141): grouped dots, neighbour offset, neighbour coordinates
        self.corr_n_groups = corr_n_groups
        self.corr_add_neighbor_offset = bool(corr_add_neighbor_offset)
        self.corr_add_neighbor_xyz = bool(corr_add_neighbor_xyz)
        self.corr_width = corr_n_groups + 3 * int(self.corr_add_neighbor_offset) + 3 * int(self.corr_add_neighbor_xyz)
        self.num_heads = num_heads
        self.dim_head = 48
        self.hidden = hidden_size
        self.depth = time_depth
        self.nv = num_virtual_tracks
        self.use_flash_attention = use_flash_attention  # accepted for config compatibility; same math
        self.updateformer_input_dim = (self.flow_embed_dim + 1) * 3 + corr_neighbors * corr_n_levels * self.corr_width + fmaps_dim + 2
        self.out_dim = 3 + fmaps_dim
        for key, shape in self._shapes().items():
            _insert(self, key, self._init_tensor(key, shape))
        self._packed: Optional[dict] = None
        self._packed_sig = None
        self._plist = None
        self._plist_epoch = -1
        self._slot_cache = {}
        # arithmetic of the matrix-core kernels (convs + linears); everything else is always fp32:
        #   "fp32"   v_mfma_f32_32x32x2_f32, exact fp32 FMA chains
        #   "bf16x3" split-precision bf16 MFMA (hi*hi + hi*lo + lo*hi), fp32-grade results, ~5x the fp32 MFMA rate
        #   "bf16"   operands rounded to bf16, fp32 accumulate (the arithmetic of torch autocast in the reference demo)
        self.precision = os.environ.get("MVT_PRECISION", "fp32")
        self.fuse_mlp = True
        self.fuse_blocks = True
        self.mfma_attention = True
        self.overlap_encoder = os.environ.get("MVT_OVERLAP", "1") != "0"  # encode later frames on a second stream
        self._side = {}
        self._scratch = {}
        self.bf16_tokens = os.environ.get("MVT_BF16_TOK", "1") != "0"  # bf16 mode: q/k/v and attention outputs stored as bf16
        self.bf16_activations = os.environ.get("MVT_BF16_ACT", "1") != "0"  # bf16 mode: encoder activations stored as bf16
        self.bf16_store = os.environ.get("MVT_BF16_STORE", "1") != "0"  # bf16 mode: bf16 feature rows in the frame store
        # attention inside the block kernels: bit 0 time, bit 1 point<-virtual, bit 2 virtual self; bit 4: the virtual<-point block
        # combines the key-split partials in its prologue, no merge launch; bit 5: the virtual-self block's pass 2 inside the
        # point<-virtual block.  Bits 4 / 5 are bit-identical to the launches they replace; bits 0-2 agree with the separate attention
        # launches to bf16 rounding only (single-pass softmax, block-diagonal time attention: round 3)
        self.fuse_attention = int(os.environ.get("MVT_FUSE_ATTN", "55"))
        self.seed_across_windows = os.environ.get("MVT_SEED_WINDOWS", "1") != "0"  # previous window's neighbours seed the first scan
        self.encoder_chunk_images = int(os.environ.get("MVT_ENC_CHUNK", "0"))  # images per encoder call (0: max(16, V * S/2))
        self.encoder_streams = int(os.environ.get("MVT_ENC_STREAMS", "2"))  # 2: the chunks of an encoder call alternate between two streams
        self.presearch = os.environ.get("MVT_PRESEARCH", "1") != "0"  # first searches of new tracks beside the first encoder block
        self.stem_reads_clip = os.environ.get("MVT_STEM_RGB", "1") != "0"  # composite encoder: the stem reads the planar clip itself
        # the wide conv2 as one 512-thread workgroup per CU (conv3x3_big_bf16) also for the encoder blocks that run on the second
        # stream BESIDE the refinement windows: "0" never (its ~100-us workgroups starve the windows' kernels -- above all the
        # latency-bound correlation gather -- of CU slots), "1" always, "auto": when the encoder, not the windows, bounds the call
        # (the second stream has more than twice the images of the first block; BASELINE config C5: -6 % per step)
        self.wide_conv_shared = os.environ.get("MVT_CONV_BIG_SHARED", "auto")
        self._shared_gpu = False  # set around the encoder calls issued beside the windows
        self.side_after_corr = os.environ.get("MVT_SIDE_AFTER_CORR", "1") != "0"  # second-stream encoder starts behind the first correlation
        # MVT_SYNC_DEBUG=1: synchronise the whole device at every cross-stream hand-over of a call ("hand-over
        # table"): if results change with it, an event / wait_stream is missing somewhere (tests: bit-identical with and without)
        self.sync_debug = os.environ.get("MVT_SYNC_DEBUG", "0") != "0"
        self.defer_encoder = os.environ.get("MVT_ENC_DEFER", "0") != "0"  # one block of later frames per window on the side stream (A/B: no gain at C3)
        self.knn_one_launch = os.environ.get("MVT_KNN_ONE_LAUNCH", "1") != "0"  # seeded scans: one wave per (track, frame), no merge launch
        self.composite_encoder = os.environ.get("MVT_COMPOSITE_ENCODER", "1") != "0"  # the CNN as one library call (bf16 mode)
        self.fuse_tokens = os.environ.get("MVT_FUSE_TOKENS", "1") != "0"  # ... with the token rows assembled inside that launch
        self.fuse_input = os.environ.get("MVT_FUSE_INPUT", "1") != "0"  # input transform + virtual tokens + first q|k|v in one launch
        self.fuse_head = os.environ.get("MVT_FUSE_HEAD", "1") != "0"  # flow head + track / feature update in one kernel
        self.fuse_norm = True  # InstanceNorm statistics from the conv epilogue + normalise-on-load (bf16 / bf16x3 convs)
        self.fold_downsample = os.environ.get("MVT_FOLD_DOWNSAMPLE", "1") != "0"  # strided blocks: conv1 + downsample[0] in one launch
        self.fuse_ln = False
        d = self.updateformer_input_dim
        self._time_embed_host = self._make_time_embed(self.S, d)

    # ------------------------------------------------------------------ parameters
    def _shapes(self) -> Dict[str, Tuple[int, ...]]:
        """state_dict contract."""
        s: Dict[str, Tuple[int, ...]] = {}

        def wb(name, *shape):
            s[name + ".weight"] = tuple(shape)
            s[name + ".bias"] = (shape[0],)

        wb("fnet.conv1", 64, 3, 7, 7)
        cin = 64
        for li, cout in ((1, 64), (2, 96), (3, 128), (4, 128)):
            wb(f"fnet.layer{li}.0.conv1", cout, cin, 3, 3)
            wb(f"fnet.layer{li}.0.conv2", cout, cout, 3, 3)
            if li != 1:
                wb(f"fnet.layer{li}.0.downsample.0", cout, cin, 1, 1)
            wb(f"fnet.layer{li}.1.conv1", cout, cout, 3, 3)
            wb(f"fnet.layer{li}.1.conv2", cout, cout, 3, 3)
            cin = cout
        wb("fnet.conv2", self.latent_dim * 2, 416, 3, 3)
        wb("fnet.conv3", self.latent_dim, self.latent_dim * 2, 1, 1)
        h, inner, mlp = self.hidden, self.num_heads * self.dim_head, int(self.hidden * 4.0)
        u = "updateformer."
        s[u + "virual_tracks"] = (1, self.nv, 1, h)
        wb(u + "input_transform", h, self.updateformer_input_dim)
        wb(u + "flow_head.0", self.out_dim, h)
        wb(u + "flow_head.2", self.out_dim, self.out_dim)
        wb(u + "flow_head.4", self.out_dim, self.out_dim)
        for i in range(self.depth):
            for blk, attn in (("time_blocks", "attn"), ("space_virtual_blocks", "attn"),
                              ("space_point2virtual_blocks", "cross_attn"), ("space_virtual2point_blocks", "cross_attn")):
                p = f"{u}{blk}.{i}"
                if attn == "cross_attn":
                    s[p + ".norm_context.weight"] = (h,)
                    s[p + ".norm_context.bias"] = (h,)
                wb(f"{p}.{attn}.to_q", inner, h)
                wb(f"{p}.{attn}.to_kv", 2 * inner, h)
                wb(f"{p}.{attn}.to_out", h, inner)
                wb(p + ".mlp.fc1", mlp, h)
                wb(p + ".mlp.fc2", h, mlp)
        s["ffeats_norm.weight"] = (self.latent_dim,)
        s["ffeats_norm.bias"] = (self.latent_dim,)
        wb("ffeats_updater.0", self.latent_dim, self.latent_dim)
        wb("vis_predictor.0", 1, self.latent_dim)
        return s

    @staticmethod
    def _init_tensor(key: str, shape) -> torch.Tensor:
        t = torch.zeros(shape)
        if key.endswith("virual_tracks"):
            return torch.randn(shape)
        if key.endswith(".bias"):
            return t
        if "norm" in key.split(".")[-2]:
            return torch.ones(shape)
        if key.startswith("fnet."):
            nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu")
        elif "flow_head" in key:
            nn.init.trunc_normal_(t, std=0.001)
        else:
            nn.init.xavier_uniform_(t)
        return t

    @staticmethod
    def _make_time_embed(S: int, D: int) -> torch.Tensor:
        """1-D sin|cos embedding of s/S, fp64 on the host, first D of D+D%2 columns (mvtracker.py:333-344)."""
        dim = D + (D % 2)
        omega = np.arange(dim // 2, dtype=np.float64)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        pos = (torch.linspace(0, S - 1, S).reshape(S, 1) / S).numpy().reshape(-1)
        out = np.einsum("m,d->md", pos, omega)
        return torch.from_numpy(np.concatenate([np.sin(out), np.cos(out)], axis=1)).float()[:, :D].contiguous()

    def register_parameter(self, name, param):
        _param_epoch[0] += 1
        super().register_parameter(name, param)

    def init_stats(self):  # reference API (mvtracker.py:190-242); statistics are not collected here
        pass

    def consume_stats(self):
        pass

    # ------------------------------------------------------------------ weight packing for the kernels
    def _signature(self, dev):
        assert self.precision in ("fp32", "bf16x3", "bf16"), self.precision
        # (everything _pack reads: the switches that decide which structs / layouts are built, then the parameter versions)
        flags = (str(dev), self.precision, self.fuse_attention, self.fuse_input, self.composite_encoder, self._composite_updater_ok(),
                 self._composite_encoder_ok(), self.bf16_tokens, self.bf16_activations, self.fuse_blocks, self.mfma_attention, self.fuse_norm,
                 self.depth, self.hidden)
        # (the Parameter objects are fixed at construction -- load_state_dict / .to() change their data in place -- so the module
        #  tree is walked once: nn.Module.parameters() costs ~0.8 ms per walk, and this runs several times per call)
        #  again only after a Parameter object has been registered somewhere: _Node.register_parameter / register_parameter below)
        if self._plist is None or self._plist_epoch != _param_epoch[0]:
            self._plist = list(self.parameters())
            self._plist_epoch = _param_epoch[0]
        return flags + tuple(p._version for p in self._plist) + tuple(p.data_ptr() for p in self._plist)

    def _pack(self, dev) -> dict:
        sig = self._signature(dev)
        if self._packed is not None and self._packed_sig == sig:
            return self._packed
        sd = {k: v.detach().to(device=dev, dtype=torch.float32) for k, v in self.state_dict().items()}
        pk: dict = {}

        prec = self.precision

        def matrix(w2d):
            """[N][K] fp32 -> zero-padded [N][round_up(K,64)] in the layout of the selected precision:
            fp32 tensor, or (bf16 hi, bf16 lo | None) as int16 tensors."""
            n, k = w2d.shape
            wp = torch.zeros(n, _round_up(k, 64), device=dev)
            wp[:, :k] = w2d
            if prec == "fp32":
                return wp
            hi = torch.empty(wp.shape, device=dev, dtype=torch.int16)
            lo = torch.empty(wp.shape, device=dev, dtype=torch.int16) if prec == "bf16x3" else None
            hip.split_bf16(wp, hi, lo, wp.numel())
            return (hi, lo)

        def conv(name):
            w = sd[name + ".weight"]
            pk[name] = (matrix(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)), sd[name + ".bias"].contiguous())

        def lin(name, w=None, b=None):
            w = sd[name + ".weight"] if w is None else w
            b = sd[name + ".bias"] if b is None else b
            n, k = w.shape
            pk[name] = (matrix(w), b.contiguous(), n, k)
            if prec == "bf16" and name.startswith("updateformer.") and k % 16 == 0 and "flow_head" not in name:
                hi = pk[name][0][0]  # fragment-major copy for the fused block kernel
                fr = torch.empty(_round_up(n, 32) * k, device=dev, dtype=torch.int16)
                hip.pack_frag_bf16(hi, hi.shape[1], n, k, fr)
                pk[name + "#frag"] = fr

        w = sd["fnet.conv1.weight"]  # (64,3,7,7) -> [64][7][32] with element kw*4+c
        st = torch.zeros(64, 7, 8, 4, device=dev)
        st[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        pk["fnet.conv1"] = (matrix(st.reshape(64, 7 * 32)), sd["fnet.conv1.bias"].contiguous())
        for k in sd:
            if k.startswith("fnet.") and k.endswith(".weight") and k != "fnet.conv1.weight":
                conv(k[:-7])
        u = "updateformer."
        for name in ("input_transform", "flow_head.0", "flow_head.2", "flow_head.4"):
            lin(u + name)
        for i in range(self.depth):
            for blk in ("time_blocks", "space_virtual_blocks"):
                p = f"{u}{blk}.{i}"
                lin(p + ".attn.qkv", torch.cat([sd[p + ".attn.to_q.weight"], sd[p + ".attn.to_kv.weight"]], 0),
                    torch.cat([sd[p + ".attn.to_q.bias"], sd[p + ".attn.to_kv.bias"]], 0))
                lin(p + ".attn.to_out")
                lin(p + ".mlp.fc1")
                lin(p + ".mlp.fc2")
            for blk in ("space_point2virtual_blocks", "space_virtual2point_blocks"):
                p = f"{u}{blk}.{i}"
                pk[p + ".norm_context"] = (sd[p + ".norm_context.weight"].contiguous(), sd[p + ".norm_context.bias"].contiguous())
                for nme in (".cross_attn.to_q", ".cross_attn.to_kv", ".cross_attn.to_out", ".mlp.fc1", ".mlp.fc2"):
                    lin(p + nme)
        pk["virtual"] = sd[u + "virual_tracks"].reshape(self.nv, self.hidden).contiguous()
        if self._composite_updater_ok():
            pk["updater_struct"] = self._updater_struct(pk, sd, matrix)
        if self._composite_encoder_ok():
            ew = hip.EncoderWeights()
            ew.latent_dim = self.latent_dim
            names = ["fnet.conv1"]
            for li in range(1, 5):
                names += [f"fnet.layer{li}.0.conv1", f"fnet.layer{li}.0.conv2", f"fnet.layer{li}.0.downsample.0" if li > 1 else f"fnet.layer{li}.0.conv1",
                          f"fnet.layer{li}.1.conv1", f"fnet.layer{li}.1.conv2"]
            names += ["fnet.conv2", "fnet.conv3"]
            ews = hip.EncoderWeights()  # the same weights for calls that share the GPU with the refinement windows (MVT_IO_SHORT_WG)
            ews.latent_dim, ews.short_workgroups = self.latent_dim, 1
            for i, nm in enumerate(names):  # (layer 1 has no downsample conv: slot 3 repeats a valid pointer and is never used)
                ew.conv[i].w, ew.conv[i].b = pk[nm][0][0].data_ptr(), pk[nm][1].data_ptr()
                ews.conv[i].w, ews.conv[i].b = ew.conv[i].w, ew.conv[i].b
            pk["encoder_struct"], pk["encoder_struct_shared"] = ew, ews
        pk["ffeats_norm"] = (sd["ffeats_norm.weight"].contiguous(), sd["ffeats_norm.bias"].contiguous())
        lin("ffeats_updater.0")
        pk["vis"] = (sd["vis_predictor.0.weight"].reshape(-1).contiguous(), sd["vis_predictor.0.bias"].contiguous())
        pk["time_embed"] = self._time_embed_host.to(dev)
        a3 = _round_up(self.updateformer_input_dim, 6) // 3  # embeddings.py:95-97 with embed_dim = dim_padded / 3
        om = np.arange(a3 // 2, dtype=np.float64)
        om /= a3 / 2.0
        pk["pos_omega"] = torch.from_numpy(1.0 / 10000 ** om).to(dev)
        self._packed, self._packed_sig = pk, sig
        return pk

    def _composite_updater_ok(self):
        """mvt_updateformer_forward covers the shipped geometry in bf16 mode with bf16 q/k/v tensors."""
        return (self.precision == "bf16" and self.hidden == 256 and self.num_heads == 6 and self.dim_head == 48 and self.nv == 64
                and self.latent_dim == 128 and self.fuse_blocks and self.mfma_attention and self.bf16_tokens
                and self.depth <= hip.UPDATER_MAX_DEPTH and hip.COMPOSITE and os.environ.get("MVT_COMPOSITE", "1") != "0")

    def _composite_encoder_ok(self):
        """mvt_encoder_forward covers bf16 mode with bf16 activations and the fused InstanceNorm path."""
        return (self.precision == "bf16" and self.bf16_activations and self.fuse_norm and self.latent_dim % 32 == 0 and hip.COMPOSITE
                and self.composite_encoder)

    def _updater_struct(self, pk, sd, matrix):
        """mvt_updater_weights (host struct of device pointers into ``pk``) for mvt_updateformer_forward."""
        u = "updateformer."
        h, inner = self.hidden, self.num_heads * self.dim_head
        w = hip.UpdaterWeights()
        w.depth, w.hidden, w.heads, w.dim_head, w.n_virtual, w.S = self.depth, h, self.num_heads, self.dim_head, self.nv, self.S
        w.token_dim, w.out_dim = self.updateformer_input_dim, self.out_dim
        w.fuse_attention = self.fuse_attention
        w.virtual_tokens = pk["virtual"].data_ptr()
        keep = []  # tensors referenced by the struct only

        def rows(name, pad_n=False):
            wt, b = sd[name + ".weight"], sd[name + ".bias"]
            n, k = wt.shape
            if pad_n and n % 4:  # zero rows + zero bias: the GEMM itself writes the pad columns of the hidden activations as zeros
                n4 = _round_up(n, 4)
                wt = torch.cat([wt, torch.zeros(n4 - n, k, device=wt.device)], 0)
                b = torch.cat([b, torch.zeros(n4 - n, device=b.device)], 0)
            hi = matrix(wt)[0]
            b = b.contiguous()
            keep.extend([hi, b])
            return hip.lin_rows(hi, b, wt.shape[0], k)

        w.input_transform = rows(u + "input_transform")
        w.flow0, w.flow2, w.flow4 = rows(u + "flow_head.0", True), rows(u + "flow_head.2", True), rows(u + "flow_head.4")

        def frag_of(name, kpad):  # fragment-major bf16 of a [N][K] linear, K zero-padded to kpad (a multiple of 16)
            wt, b = sd[name + ".weight"], sd[name + ".bias"].contiguous()
            n, k = wt.shape
            if kpad > _round_up(k, 64):  # (the row-major image must be at least kpad wide: token widths well below 592)
                wt = torch.cat([wt, torch.zeros(n, kpad - k, device=wt.device)], 1)
            hi = matrix(wt)[0]  # [n][round_up(k, 64)], zero padded
            fr = torch.empty(_round_up(n, 32) * kpad, device=wt.device, dtype=torch.int16)
            hip.pack_frag_bf16(hi, hi.shape[1], n, kpad, fr)
            keep.extend([fr, b])
            return hip.lin_frag(fr, b, n, kpad)

        w.flow0_frag, w.flow2_frag, w.flow4_frag = frag_of(u + "flow_head.0", h), frag_of(u + "flow_head.2", 144), frag_of(u + "flow_head.4", 144)
        w.ffeats_updater = frag_of("ffeats_updater.0", self.latent_dim)
        if self.fuse_input and self.updateformer_input_dim <= 592:
            w.input_frag = frag_of(u + "input_transform", 592)
        gw, gb = sd["ffeats_norm.weight"].contiguous(), sd["ffeats_norm.bias"].contiguous()
        keep.extend([gw, gb])
        w.ffeats_norm_w, w.ffeats_norm_b = gw.data_ptr(), gb.data_ptr()

        def frag(name):
            _, b, n, k = pk[name]
            return hip.lin_frag(pk[name + "#frag"], b, n, k)

        for i in range(self.depth):
            for arr, blk, attn, cross in ((w.time_blk, "time_blocks", "attn", False), (w.vself, "space_virtual_blocks", "attn", False),
                                          (w.v2p, "space_virtual2point_blocks", "cross_attn", True),
                                          (w.p2v, "space_point2virtual_blocks", "cross_attn", True)):
                p = f"{u}{blk}.{i}"
                b = arr[i]
                if cross:
                    b.q, b.kv = frag(f"{p}.{attn}.to_q"), frag(f"{p}.{attn}.to_kv")
                    b.ctx_ln_w, b.ctx_ln_b = (t_.data_ptr() for t_ in pk[p + ".norm_context"])
                else:
                    b.qkv = frag(f"{p}.{attn}.qkv")
                b.out, b.fc1, b.fc2 = frag(f"{p}.{attn}.to_out"), frag(p + ".mlp.fc1"), frag(p + ".mlp.fc2")
        pk["updater_struct_keep"] = keep
        return w

    # ------------------------------------------------------------------ encoder (reference spatracker/blocks.py:214-284)
    def _conv(self, pk, name, x, n, H, W, cin, cout, k, stride, pad, out=None, ldo=None, in_stats=None, stats=False):
        """Convolution; ``stats=True`` also returns the InstanceNorm (mean, rstd) of the output (taken from the conv epilogue
        on the bf16 kernels), ``in_stats`` makes the kernel read its input as relu(IN(x)) (bf16 3x3 stride-1 only)."""
        wt, b = pk[name]
        Ho = (H + 2 * pad - k) // stride + 1
        Wo = (W + 2 * pad - k) // stride + 1
        if out is None:
            out = torch.empty(n, Ho, Wo, cout, device=x.device, dtype=self._act_dtype(pk))
            ldo = cout
        st = None
        if isinstance(wt, tuple):
            slots = hip.conv2d_stat_slots(H, W, cin, k, k, stride, pad, wt[1] is not None) if (stats and self.fuse_norm) else 0
            part = torch.empty(n * slots * cout * 2, device=x.device) if slots else None
            hip.conv2d_bf16(x, wt[0], wt[1], b, out, n, H, W, cin, cout, k, k, stride, pad, ldo, in_stats=in_stats, out_partial=part,
                            short_wg=self._shared_gpu)
            if slots:
                st = torch.empty(n, cout, 2, device=x.device)
                hip.instnorm_finish_slots(part, slots, st, n, Ho * Wo, cout)
        else:
            assert in_stats is None
            hip.conv2d(x, wt, b, out, n, H, W, cin, cout, k, k, stride, pad, ldo)
        if stats and st is None:
            st = self._inorm(out, n, Ho * Wo, cout, apply=False)
        return (out, Ho, Wo, st) if stats else (out, Ho, Wo)

    def store_dtype(self):
        """Element type of the frame store's feature rows: bf16 in bf16 mode -- under autocast the reference's fmaps / pc_fvec are
        bf16 tensors (model_utils.py:467-476) -- which halves the store, the correlation kernel's gather traffic and the
        multi-GPU all-gather; fp32 in the fp32 / bf16x3 modes."""
        return torch.bfloat16 if (self.precision == "bf16" and self.bf16_store) else torch.float32

    def _act_dtype(self, pk):
        """Element type of the encoder's intermediate activations: bf16 in bf16 mode (HBM-bound layers, bf16 MFMA operands)."""
        return torch.bfloat16 if (self.bf16_activations and self.precision == "bf16") else torch.float32

    def _inorm(self, x, n, HW, C, skip=None, skip_stats=None, apply=True, st=None, skip_relu=False):
        if st is None:
            partial = torch.empty(n * hip.IN_SLABS * C * 2, device=x.device, dtype=torch.float64)
            st = torch.empty(n, C, 2, device=x.device)
            hip.instnorm_stats(x, C, partial, st, n, HW, C)
        if apply:
            hip.instnorm_apply(x, st, skip, skip_stats, x, n, HW, C, skip_relu=skip_relu)
        return st

    def _res_block(self, pk, p, x, n, H, W, cin, cout, stride, x_stats=None):
        """ResidualBlock (blocks.py:84-128).  ``x_stats``: x is a raw conv output whose InstanceNorm + ReLU was never
        materialised (the stem): conv1 normalises while it loads and the skip connection is normalised in the final pass."""
        fuse_in = self.fuse_norm and isinstance(pk[p + ".conv2"][0], tuple) and cout % 32 == 0
        assert x_stats is None or (stride == 1 and (p + ".downsample.0") not in pk)
        d = dst = None
        fold = (fuse_in and stride == 2 and (p + ".downsample.0") in pk and x.dtype == torch.bfloat16 and pk[p + ".conv1"][0][1] is None
                and self.fold_downsample)
        if fold:
            # conv1 (3x3 / stride 2) and downsample[0] (1x1 / stride 2) read the same x: one launch (mvt_conv3x3s2_down_bf16)
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            y = torch.empty(n, Ho, Wo, cout, device=x.device, dtype=torch.bfloat16)
            d = torch.empty(n, Ho, Wo, cout, device=x.device, dtype=torch.bfloat16)
            slots = hip.conv2d_stat_slots(H, W, cin, 3, 3, 2, 1, False)
            part = torch.empty(2, n * slots * cout * 2, device=x.device)
            (w1, b1), (wd, bd) = pk[p + ".conv1"], pk[p + ".downsample.0"]
            hip.conv3x3s2_down_bf16(x, w1[0], b1, wd[0], bd, y, d, n, H, W, cin, cout, cout, part[0], part[1])
            st1, dst = torch.empty(n, cout, 2, device=x.device), torch.empty(n, cout, 2, device=x.device)
            hip.instnorm_finish_slots(part[0], slots, st1, n, Ho * Wo, cout)
            hip.instnorm_finish_slots(part[1], slots, dst, n, Ho * Wo, cout)
        else:
            y, Ho, Wo, st1 = self._conv(pk, p + ".conv1", x, n, H, W, cin, cout, 3, stride, 1, stats=True, in_stats=x_stats)
        if not fuse_in:  # otherwise conv2 normalises while it loads its patch
            self._inorm(y, n, Ho * Wo, cout, st=st1)
        y2, _, _, st2 = self._conv(pk, p + ".conv2", y, n, Ho, Wo, cout, cout, 3, 1, 1, in_stats=st1 if fuse_in else None,
                                   stats=True)
        if (p + ".downsample.0") in pk:
            if d is None:
                d, _, _, dst = self._conv(pk, p + ".downsample.0", x, n, H, W, cin, cout, 1, stride, 0, stats=True)
            self._inorm(y2, n, Ho * Wo, cout, skip=d, skip_stats=dst, st=st2)
        else:
            self._inorm(y2, n, Ho * Wo, cout, skip=x, skip_stats=x_stats, skip_relu=x_stats is not None, st=st2)
        return y2, Ho, Wo

    def _encode(self, pk, x4, n, H, W, out_rows):
        """x4 (n,H,W,4) normalised RGB -- or a ``_ClipImages`` reference to n images of the planar clip, which the composite
        encoder's stem reads directly -- -> writes (n, H/4, W/4, C) into ``out_rows``."""
        C = self.latent_dim
        composite = "encoder_struct" in pk and out_rows.is_contiguous()
        est = "encoder_struct_shared" if self._shared_gpu else "encoder_struct"
        if isinstance(x4, _ClipImages):
            src = x4
            if composite and self.stem_reads_clip:  # mvt_encoder_forward_rgb: no (n,H,W,4) staging tensor, one launch less
                ws = self._workspace(hip.encoder_workspace_bytes(n, H, W, C), src.rgbs.device)
                return hip.encoder_forward_rgb(pk[est], src.rgbs, src.V, src.T, src.img0, n, H, W, out_rows, C, ws)
            x4 = torch.empty(n, H, W, 4, device=src.rgbs.device)
            hip.rgb_images_to_nhwc4(src.rgbs, x4, src.V, src.T, H, W, src.img0, n)
        if composite:  # the whole CNN as ONE library call (mvt_encoder_forward)
            ws = self._workspace(hip.encoder_workspace_bytes(n, H, W, C), x4.device)
            return hip.encoder_forward(pk[est], x4, n, H, W, out_rows, C, ws)
        hs, ws = H // self.stride, W // self.stride
        x, h, w, st = self._conv(pk, "fnet.conv1", x4, n, H, W, 4, 64, 7, 2, 3, stats=True)
        lazy_stem = self.fuse_norm and self.precision == "bf16"  # relu(IN(stem)) is applied by its two consumers instead
        if not lazy_stem:
            self._inorm(x, n, h * w, 64, st=st)
        cat = torch.empty(n, hs, ws, 416, device=x4.device, dtype=self._act_dtype(pk))
        cin = 64
        stages, dims = [], []
        for li, (cout, stride) in enumerate(((64, 1), (96, 2), (128, 2), (128, 2)), start=1):
            x, h, w = self._res_block(pk, f"fnet.layer{li}.0", x, n, h, w, cin, cout, stride, x_stats=st if (li == 1 and lazy_stem) else None)
            x, h, w = self._res_block(pk, f"fnet.layer{li}.1", x, n, h, w, cout, cout, 1)
            stages.append(x)
            dims.append((h, w, cout))
            cin = cout
        # the four resized stage outputs side by side (blocks.py:266-276), one launch writing whole 416-channel rows
        hip.concat_resize_bilinear_ac(stages, dims, cat, n, hs, ws, 416)
        y, _, _, st = self._conv(pk, "fnet.conv2", cat, n, hs, ws, 416, 2 * C, 3, 1, 1, stats=True)
        if self.fuse_norm and self.precision == "bf16" and (2 * C) % 32 == 0:
            # the 1x1 output conv normalises while it loads (row-tile kernel): no separate InstanceNorm pass
            self._conv(pk, "fnet.conv3", y, n, hs, ws, 2 * C, C, 1, 1, 0, out=out_rows, ldo=C, in_stats=st)
        else:
            self._inorm(y, n, hs * ws, 2 * C, st=st)
            self._conv(pk, "fnet.conv3", y, n, hs, ws, 2 * C, C, 1, 1, 0, out=out_rows, ldo=C)

    @hip.guarded
    def encode_images(self, rgbs, i0, i1, out, images_per_chunk=16, after_first_chunk=None):
        """Encode images [i0, i1) of rgbs (V,T,3,H,W) in [0,255] -- images numbered frame-major, t * V + v, the order of the
        frame store -- into ``out`` (>= i1 images, H/4, W/4, C): image i lands in out[i]."""
        V, T, _, H, W = rgbs.shape
        dev = rgbs.device
        pk = self._pack(dev)
        starts = list(range(i0, i1, images_per_chunk))
        # MVT_ENC_STREAMS=2: the chunks of one call alternate between the caller's stream and a helper stream, so that the small
        # layers of one chunk (layer 3 / 4 and the 1x1 convolutions fill a fraction of the chip) run beside the big layers of the other
        two = self.encoder_streams == 2 and dev.type == "cuda" and len(starts) > 1
        cur = torch.cuda.current_stream(dev) if two else None
        helper = self._helper_stream(dev) if two else None
        if two:
            helper.wait_stream(cur)  # hand-over H1: the helper stream reads the clip / writes store rows the caller's stream owns
            self._handover(dev)
        joins = []
        for ci, a in enumerate(starts):
            n = min(images_per_chunk, i1 - a)
            on_helper = two and ci % 2 == 1
            ctx = torch.cuda.stream(helper) if on_helper else contextlib.nullcontext()
            with ctx:
                # (images a .. a+n-1 of the clip by reference: the composite encoder's stem reads the planar frames itself)
                self._encode(pk, _ClipImages(rgbs, V, T, a), n, H, W, out[a:a + n])
                if on_helper:
                    ev = torch.cuda.Event()
                    ev.record(helper)
                    joins.append(ev)
            if after_first_chunk is not None:  # (host-side hook: the GPU has work queued now, see MVTracker.forward)
                after_first_chunk()
                after_first_chunk = None
        for ev in joins:
            cur.wait_event(ev)  # hand-over H2: the helper's chunks (store rows) back to the caller's stream
        if joins:
            self._handover(dev)
        if after_first_chunk is not None:
            after_first_chunk()

    def encode_frames(self, rgbs, t0=0, t1=None, images_per_chunk=16, out=None, out_t0=0, after_first_chunk=None):
        """rgbs (V,T,3,H,W) in [0,255] -> level-0 features (T,V,H/4,W/4,C); frames outside [t0,t1) are left zero
        (or untouched when the result goes into ``out``, whose first row is frame ``out_t0``)."""
        V, T, _, H, W = rgbs.shape
        t1 = T if t1 is None else t1
        hs, ws = H // self.stride, W // self.stride
        F0 = out if out is not None else torch.zeros(T, V, hs, ws, self.latent_dim, device=rgbs.device, dtype=self.store_dtype())
        flat = F0.view(-1, hs, ws, self.latent_dim)
        # (whole frames per chunk, as before: the chunk boundaries do not change any result, only the launch shapes)
        self.encode_images(rgbs, t0 * V, t1 * V, _Shifted(flat, out_t0 * V), images_per_chunk=max(1, images_per_chunk // V) * V,
                           after_first_chunk=after_first_chunk)
        return F0

    def _pinned_i64(self, dev, n):
        """Cached pinned host buffer of n int64 (per device and stream) for the asynchronous read-back of the query frames."""
        key = ("pin64", dev.index, torch.cuda.current_stream(dev).cuda_stream)
        t = self._scratch.get(key)
        if t is None or t.numel() < n:
            t = self._scratch[key] = torch.empty(max(n, 1024), dtype=torch.int64).pin_memory()
        return t[:n]

    def _upload_small(self, dev, *arrays):
        """Host numpy arrays -> device tensors via one cached pinned buffer (non-blocking copies on the current stream)."""
        if dev.type != "cuda":
            return tuple(torch.from_numpy(a).to(dev) for a in arrays)
        nbytes = sum((a.nbytes + 15) // 16 * 16 for a in arrays)
        key = ("pin", dev.index, torch.cuda.current_stream(dev).cuda_stream)  # (per stream: single_point mode runs forwards on several)
        pin = self._scratch.get(key)
        if pin is None or pin.numel() < nbytes:
            pin = self._scratch[key] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        out, o = [], 0
        for a in arrays:
            view = pin[o:o + a.nbytes].view(torch.from_numpy(a).dtype)
            view.copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(-1))
            out.append(view.to(dev, non_blocking=True).reshape(a.shape))
            o += (a.nbytes + 15) // 16 * 16
        return tuple(out)

    def _handover(self, dev):
        """Debug hook at every point where one stream starts consuming what another produced (see ``sync_debug``)."""
        if self.sync_debug and dev.type == "cuda":
            torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ frame store (model_utils.py:420-482)
    def _helper_stream(self, dev):
        key = ("helper", dev.type, dev.index)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=dev)
        return self._side[key]

    def _side_stream(self, dev):
        key = (dev.type, dev.index)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=dev)
        return self._side[key]

    def _encode_on_side_stream(self, store, rgbs, firsts, pending):
        """Encode the S/2-frame blocks starting at ``firsts`` on the second HIP stream (ordered after everything enqueued on the
        caller's stream so far); ``pending`` receives (first frame, event) per block."""
        dev = rgbs.device
        T, S = rgbs.shape[1], self.S
        # (the later blocks stay on ONE stream, in order: alternating them between two streams as encode_images does for the chunks
        #  of the first block measured +0.5 ms -- beside the updater their concurrency only adds contention)
        streams = [self._side_stream(dev)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))  # hand-over H3: the store (allocated / geometry written on the caller's stream)
        self._handover(dev)
        # these blocks run BESIDE the refinement windows: short-lived workgroups only, unless the encoder bounds the call anyway
        V = rgbs.shape[0]
        first_block = max(1, min(T, S)) * V
        side_images = sum(min(T, a + S // 2) - a for a in firsts) * V
        big = self.wide_conv_shared == "1" or (self.wide_conv_shared == "auto" and side_images > 2 * first_block)
        for i, a in enumerate(firsts):
            st = streams[i % len(streams)]
            with torch.cuda.stream(st):
                self._shared_gpu = not big
                try:
                    self.fill_frame_features(store, rgbs, a, min(T, a + S // 2))
                finally:
                    self._shared_gpu = False
                ev = torch.cuda.Event()
                ev.record(st)
                pending.append((a, ev))

    @hip.guarded
    def fill_frame_features(self, store, rgbs, a, b, level0=None, after_first_chunk=None):
        """Encode frames [a, b) into the store's feature pyramid (everything else in the store is geometry)."""
        V, T, _, H, W = rgbs.shape
        hs, ws = H // self.stride, W // self.stride
        fv = store["fvec"]
        if level0 is None:
            self.encode_frames(rgbs, a, b, images_per_chunk=self.encoder_chunk_images or max(16, V * (self.S // 2)), out=fv[0],
                               after_first_chunk=after_first_chunk)
        elif after_first_chunk is not None:
            after_first_chunk()
        for lvl in range(1, self.corr_n_levels):
            h, w = hs >> (lvl - 1), ws >> (lvl - 1)
            hip.avgpool2(fv[lvl - 1][a:b], fv[lvl][a:b], (b - a) * V, h, w, self.latent_dim)

    @hip.guarded
    def store_geometry(self, depths, intrs, extrs):
        """The query-independent half of the frame store: world-space points of every pyramid level, tile boxes.  A handful of small
        kernels; ``forward`` enqueues them BEFORE its one host sync (the query frames), so that the GPU has work while the host wakes up."""
        V, T, _, H, W = depths.shape
        dev = depths.device
        hs, ws = H // self.stride, W // self.stride
        # (the reference fails here too, inside its kNN -- pointops / topk with k above the number of points, mvtracker.py:26-90 --; say why)
        pmin = V * (hs >> (self.corr_n_levels - 1)) * (ws >> (self.corr_n_levels - 1))
        if pmin < self.corr_neighbors:
            raise ValueError(f"the coarsest level of the point-cloud pyramid has {pmin} points per frame ({V} views of "
                             f"{hs >> (self.corr_n_levels - 1)} x {ws >> (self.corr_n_levels - 1)}), fewer than corr_neighbors = "
                             f"{self.corr_neighbors}: frames of {H} x {W} are too small for {self.corr_n_levels} correlation levels")
        kinv = torch.empty(V * T, 9, device=dev)
        einv = torch.empty(V * T, 12, device=dev)
        hip.invert_cameras(intrs.reshape(V * T, 9), extrs.reshape(V * T, 12), kinv, einv, V * T)
        ds = torch.empty(T, V, hs, ws, device=dev)
        hip.depth_subsample(depths.reshape(V, T, H, W), ds, V, T, H, W, self.stride)
        xyz = []
        for lvl in range(self.corr_n_levels):
            o = torch.empty(T, V, hs >> lvl, ws >> lvl, 4, device=dev)
            hip.unproject(ds, kinv, einv, o, V, T, hs, ws, self.stride, lvl)
            xyz.append(o)
        P = [V * (hs >> lvl) * (ws >> lvl) for lvl in range(self.corr_n_levels)]
        # bounding boxes of the 64-point scan tiles (8x8 pixel patches where the grid allows it): the kNN scan culls with them
        # ... and the union boxes of 64 consecutive tiles, the coarse level the single-wave searches test first
        box, tgrid, gbox = [], [], []
        for lvl in range(self.corr_n_levels):
            h, w = hs >> lvl, ws >> lvl
            g = (w, h) if (w % 8 == 0 and h % 8 == 0) else (0, 0)
            nt = (P[lvl] + 63) // 64
            b = torch.empty(T, nt, 8, device=dev)
            hip.tile_aabb(xyz[lvl], P[lvl], T, b, g)
            gb = None
            if nt > 64:
                gb = torch.empty(T, (nt + 63) // 64, 8, device=dev)
                hip.tile_group_aabb(b, P[lvl], T, gb)
            box.append(b)
            tgrid.append(g)
            gbox.append(gb)
        geo = {"xyz": xyz, "P": P, "T": T, "depth_s": ds, "box": box, "tile_grid": tgrid, "gbox": gbox}
        if dev.type == "cuda":
            geo["geo_event"] = torch.cuda.Event()
            geo["geo_event"].record(torch.cuda.current_stream(dev))
        return geo

    @hip.guarded
    def build_frame_store(self, rgbs, depths, intrs, extrs, t0=0, level0=None, t1=None, after_geometry=None, geometry=None):
        """Features and world-space points of every pyramid level, frame-major.

        rgbs (V,T,3,H,W), depths (V,T,1,H,W), intrs (V,T,3,3), extrs (V,T,3,4).  ``level0`` (T,V,H/4,W/4,C)
        may carry level-0 features encoded elsewhere (frames split across GPUs, mvtracker_amd.parallel).
        Features are computed for frames [t0, t1) only (``fill_frame_features`` adds more later); the geometry
        (points, tile boxes) covers every frame."""
        V, T, _, H, W = rgbs.shape
        dev = rgbs.device
        hs, ws = H // self.stride, W // self.stride
        C = self.latent_dim
        t1 = T if t1 is None else t1
        # (no memset of the 0.8 GB level-0 store: frames [t0, T) are written by the encoder before any window reads them --
        #  later frames possibly on the second stream, ordered by events -- and frames before t0 are never read)
        sdt = level0.dtype if level0 is not None else self.store_dtype()
        fv = [torch.empty(T, V, hs, ws, C, device=dev, dtype=sdt) if level0 is None else level0]
        for lvl in range(1, self.corr_n_levels):
            fv.append(torch.empty(T, V, hs >> lvl, ws >> lvl, C, device=dev, dtype=sdt))
        for f_ in fv[(1 if level0 is not None else 0):]:
            f_[:t0].zero_()
        # geometry first (a handful of small kernels), features after: what only needs the point clouds -- the first, unseeded
        # neighbour searches of new tracks -- can then run beside the encoder.  ``after_geometry(store)`` is called on the host as
        # soon as the FIRST encoder chunk has been enqueued (the GPU is busy from then on); work it issues on another stream
        # orders itself after store["geo_event"], not after the encoder.  ``geometry``: already enqueued by the caller (store_geometry).
        store = dict(geometry) if geometry is not None else self.store_geometry(depths, intrs, extrs)
        store["fvec"] = fv
        self.fill_frame_features(store, rgbs, t0, t1, level0, after_first_chunk=(lambda: after_geometry(store)) if after_geometry else None)
        return store

    def _nseg(self, P: int, K: int) -> int:
        """Segments (runs of 64-point tiles) scanned by different waves; none may be empty."""
        nt = (P + 63) // 64
        n = max(1, min(64 // K, P // 8192, 4))
        while n > 1 and ((nt + n - 1) // n) * (n - 1) >= nt:
            n -= 1
        return n

    # ------------------------------------------------------------------ updater (cotracker2/blocks.py:455-494)
    def _lin(self, pk, name, A, lda, M, out, ldc, act=hip.ACT_NONE, R=None, ldr=0):
        wp, b, n, k = pk[name]
        if isinstance(wp, tuple):
            hip.gemm_bf16(A, lda, wp[0], wp[1], wp[0].shape[1], b, R, ldr, out, ldc, M, n, k, act)
        else:
            hip.gemm(A, lda, wp, wp.shape[1], b, R, ldr, out, ldc, M, n, k, act)

    def _ln_lin(self, pk, name, x, rows, out, ldc, scratch, ln_w=None, ln_b=None, eps=1e-6, act=hip.ACT_NONE):
        """out = act(LayerNorm(x) @ W^T + b).  mvt_ln_gemm_bf16 (LayerNorm inside the GEMM's A loader) exists but measured
        slower than the separate 7 us LayerNorm pass + GEMM (every column tile recomputes the row statistics), so it is off."""
        h = self.hidden
        wp, b, n, k = pk[name]
        if isinstance(wp, tuple) and self.fuse_ln:
            hip.ln_gemm_bf16(x, h, ln_w, ln_b, eps, wp[0], wp[1], wp[0].shape[1], b, None, 0, out, ldc, rows, n, k, act)
            return
        hip.layernorm(x, h, ln_w, ln_b, scratch, h, rows, h, eps)
        self._lin(pk, name, scratch, h, rows, out, ldc, act)

    def _mlp_residual(self, pk, p, tok, rows, xn, hbuf):
        h = self.hidden
        w1, w2 = pk[p + ".mlp.fc1"], pk[p + ".mlp.fc2"]
        if self.precision == "bf16" and h == 256 and self.fuse_mlp and rows >= 4096:  # LN + fc1 + GELU + fc2 + residual in one kernel
            hip.mlp_fused_bf16(tok, h, w1[0][0], w1[0][0].shape[1], w1[1], w2[0][0], w2[0][0].shape[1], w2[1], rows, h, 4 * h, 1e-6)
            return
        self._ln_lin(pk, p + ".mlp.fc1", tok, rows, hbuf, 4 * h, xn, act=hip.ACT_GELU_TANH)
        self._lin(pk, p + ".mlp.fc2", hbuf, 4 * h, rows, tok, h, R=tok, ldr=h)

    def _update_former(self, pk, x, ldx, n, delta, ldd, coords=None, ffeats=None, nan_flag=None):
        """delta = updater(x).  With ``coords`` / ``ffeats`` (composite path only) the track / feature update of mvtracker.py:392-399
        is applied inside the same library call and True is returned."""
        if "updater_struct" in pk:  # the whole transformer as ONE library call (mvt_updateformer_forward)
            nbytes = hip.updateformer_workspace_bytes(n, self.S)
            ws = self._workspace(nbytes, x.device)
            fused = coords is not None and self.fuse_head
            hip.updateformer_forward(pk["updater_struct"], x, ldx, n, delta, ldd, ws, coords if fused else None,
                                     ffeats if fused else None, nan_flag if fused else None)
            return fused
        if self.precision == "bf16" and self.hidden == 256 and self.num_heads * self.dim_head == 288 and self.fuse_blocks:
            return self._update_former_fused(pk, x, ldx, n, delta, ldd)
        S, h, nv, H, dh = self.S, self.hidden, self.nv, self.num_heads, self.dim_head
        inner = H * dh
        dev = x.device
        Mp, Mv = n * S, nv * S
        M = Mp + Mv
        tok = torch.empty(M, h, device=dev)
        xn = torch.empty(M, h, device=dev)
        qkv = torch.empty(M, 3 * inner, device=dev)
        att = torch.empty(M, inner, device=dev)
        hbuf = torch.empty(M, 4 * h, device=dev)
        u = "updateformer."
        self._lin(pk, u + "input_transform", x, ldx, Mp, tok, h)
        hip.broadcast_rows(pk["virtual"], tok[Mp:], h, nv, S, h)
        pt, vt = tok[:Mp], tok[Mp:]
        for i in range(self.depth):
            # time attention over the S frames of every (point or virtual) track
            p = f"{u}time_blocks.{i}"
            self._ln_lin(pk, p + ".attn.qkv", tok, M, qkv, 3 * inner, xn)
            hip.attention(qkv, 3 * inner, S, 1, qkv[:, inner:], qkv[:, 2 * inner:], 3 * inner, S, 1, att, inner, n + nv, S, S, H, dh)
            self._lin(pk, p + ".attn.to_out", att, inner, M, tok, h, R=tok, ldr=h)
            self._mlp_residual(pk, p, tok, M, xn, hbuf)
            # virtual <- point cross attention, per frame (row of item j in frame t is j*S + t)
            p = f"{u}space_virtual2point_blocks.{i}"
            self._ln_lin(pk, p + ".cross_attn.to_q", vt, Mv, qkv[Mp:], 3 * inner, xn[Mp:])
            self._ln_lin(pk, p + ".cross_attn.to_kv", pt, Mp, qkv[:Mp, inner:], 3 * inner, xn[:Mp], *pk[p + ".norm_context"], eps=1e-5)
            hip.attention(qkv[Mp:], 3 * inner, 1, S, qkv[:Mp, inner:], qkv[:Mp, 2 * inner:], 3 * inner, 1, S, att[Mp:], inner, S, nv, n, H,
                          dh)
            self._lin(pk, p + ".cross_attn.to_out", att[Mp:], inner, Mv, vt, h, R=vt, ldr=h)
            self._mlp_residual(pk, p, vt, Mv, xn[Mp:], hbuf[Mp:])
            # virtual self attention, per frame
            p = f"{u}space_virtual_blocks.{i}"
            self._ln_lin(pk, p + ".attn.qkv", vt, Mv, qkv[Mp:], 3 * inner, xn[Mp:])
            hip.attention(qkv[Mp:], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, att[Mp:], inner, S, nv, nv, H,
                          dh)
            self._lin(pk, p + ".attn.to_out", att[Mp:], inner, Mv, vt, h, R=vt, ldr=h)
            self._mlp_residual(pk, p, vt, Mv, xn[Mp:], hbuf[Mp:])
            # point <- virtual cross attention, per frame
            p = f"{u}space_point2virtual_blocks.{i}"
            self._ln_lin(pk, p + ".cross_attn.to_q", pt, Mp, qkv[:Mp], 3 * inner, xn[:Mp])
            self._ln_lin(pk, p + ".cross_attn.to_kv", vt, Mv, qkv[Mp:, inner:], 3 * inner, xn[Mp:], *pk[p + ".norm_context"], eps=1e-5)
            hip.attention(qkv[:Mp], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, att[:Mp], inner, S, n, nv, H,
                          dh)
            self._lin(pk, p + ".cross_attn.to_out", att[:Mp], inner, Mp, pt, h, R=pt, ldr=h)
            self._mlp_residual(pk, p, pt, Mp, xn[:Mp], hbuf[:Mp])
        od = self.out_dim
        ldh = _round_up(od, 4)
        h1, h2 = self._flow_scratch(Mp, ldh, dev)
        self._lin(pk, u + "flow_head.0", pt, h, Mp, h1, ldh, hip.ACT_RELU)
        self._lin(pk, u + "flow_head.2", h1, ldh, Mp, h2, ldh, hip.ACT_RELU)
        self._lin(pk, u + "flow_head.4", h2, ldh, Mp, delta, ldd)

    def _workspace(self, nbytes, dev):
        """Device scratch of the composite entry points (no state between calls; grown on demand, one per device and stream)."""
        key = ("ws", dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        t = self._scratch.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(_round_up(nbytes, 1 << 20), device=dev, dtype=torch.uint8)
            self._scratch[key] = t
        return t

    def _flow_scratch(self, rows, ld, dev):
        """Hidden activations of the flow head, (rows, ld) with ld = round_up(out_dim, 4): the GEMMs write out_dim columns, the
        pad columns are read as K padding by the next layer and must be zero -- zeroed once, then reused by every call."""
        key = (rows, ld, dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        if key not in self._scratch:
            if len(self._scratch) > 32:
                self._scratch.clear()
            self._scratch[key] = (torch.zeros(rows, ld, device=dev), torch.zeros(rows, ld, device=dev))
        return self._scratch[key]

    # ---- fused path (precision "bf16", hidden 256): per layer 4 attention launches + 5 fused block launches
    def _fused_block(self, pk, p, attn_key, x, rows, att, nexts, ws=None):
        """x += att @ Wo^T + bo; x += MLP(LN(x)); then the follow-up projections of LN(x) listed in ``nexts``."""
        h = self.hidden
        no, n1, n2 = f"{p}.{attn_key}.to_out", p + ".mlp.fc1", p + ".mlp.fc2"
        inner = self.num_heads * self.dim_head
        hip.block_fused_bf16(x, h, att, inner, inner, pk[no + "#frag"], inner, pk[no][1], pk[n1 + "#frag"], h, pk[n1][1],
                             pk[n2 + "#frag"], 4 * h, pk[n2][1], 4 * h, nexts, rows, h, ws=ws)

    def _next(self, pk, name, y, ldy, ln=None, eps=1e-6, rows=(0, 0)):
        _, b, n, k = pk[name]
        d = dict(w=pk[name + "#frag"], ldw=k, b=b, N=n, y=y, ldy=ldy, eps=eps, rows=rows)
        if ln is not None:
            d.update(lnw=ln[0], lnb=ln[1])
        return d

    def _update_former_fused(self, pk, x, ldx, n, delta, ldd):
        S, h, nv, H, dh = self.S, self.hidden, self.nv, self.num_heads, self.dim_head
        inner = H * dh
        dev = x.device
        Mp, Mv = n * S, nv * S
        M = Mp + Mv
        space_mfma = self.mfma_attention and dh == 48
        tok = torch.empty(M, h, device=dev)
        xn = torch.empty(M, h, device=dev)
        # q|k|v of the time / virtual-self attention; cross attention: q in [:, :inner], k|v in [:, inner:].  Two buffers,
        # swapped every layer: the next layer's time q|k|v is projected (by the block epilogues) while this layer's
        # cross-attention k|v are still being read.
        # (bf16 tensors in bf16 mode: they only ever feed bf16 MFMA operands, and the block / attention kernels are bound
        #  by exactly this traffic)
        tdt = torch.bfloat16 if (self.bf16_tokens and space_mfma) else torch.float32
        qkv = torch.empty(M, 3 * inner, device=dev, dtype=tdt)
        qkv_nx = torch.empty(M, 3 * inner, device=dev, dtype=tdt)
        qp = torch.empty(Mp, inner, device=dev, dtype=tdt)       # point <- virtual queries (computed right after the time block)
        att = torch.empty(M, inner, device=dev, dtype=tdt)
        ws = torch.empty(5 * Mv * h, device=dev) if 4 * h == 1024 else None  # split path of the virtual-track blocks
        aws = torch.empty(hip.attention_ws_floats(S, nv, H), device=dev) if space_mfma else None  # key-split virtual <- point attention
        u = "updateformer."
        space_attn = hip.attention_bf16 if space_mfma else hip.attention
        time_attn = hip.attention_bf16 if space_mfma else hip.attention  # 12 keys pad to one 32-key MFMA block: still 1.4x the VALU kernel
        self._lin(pk, u + "input_transform", x, ldx, Mp, tok, h)
        hip.broadcast_rows(pk["virtual"], tok[Mp:], h, nv, S, h)
        pt, vt = tok[:Mp], tok[Mp:]
        if h == 256:  # LayerNorm + projection in one launch (the split path's second pass without a workspace)
            hip.ln_proj_bf16(tok, h, [self._next(pk, f"{u}time_blocks.0.attn.qkv", qkv, 3 * inner)], M, h)
        else:
            self._ln_lin(pk, f"{u}time_blocks.0.attn.qkv", tok, M, qkv, 3 * inner, xn)
        for i in range(self.depth):
            tb, v2p = f"{u}time_blocks.{i}", f"{u}space_virtual2point_blocks.{i}"
            vs, p2v = f"{u}space_virtual_blocks.{i}", f"{u}space_point2virtual_blocks.{i}"
            last = i + 1 == self.depth
            nxt_qkv = f"{u}time_blocks.{i + 1}.attn.qkv"
            # time attention, then the rest of the time block; its epilogue already projects what the space blocks need
            time_attn(qkv, 3 * inner, S, 1, qkv[:, inner:], qkv[:, 2 * inner:], 3 * inner, S, 1, att, inner, n + nv, S, S, H, dh)
            # (one launch over point and virtual rows: same weights, the follow-up projections differ by row range)
            self._fused_block(pk, tb, "attn", tok, M, att,
                              [self._next(pk, v2p + ".cross_attn.to_kv", qkv[:, inner:], 3 * inner, pk[v2p + ".norm_context"], 1e-5,
                                          rows=(0, Mp)),
                               self._next(pk, p2v + ".cross_attn.to_q", qp, inner, rows=(0, Mp)),
                               self._next(pk, v2p + ".cross_attn.to_q", qkv, 3 * inner, rows=(Mp, M))])
            # virtual <- point
            space_attn(qkv[Mp:], 3 * inner, 1, S, qkv[:Mp, inner:], qkv[:Mp, 2 * inner:], 3 * inner, 1, S, att[Mp:], inner, S, nv, n, H,
                       dh, **({"ws": aws} if space_mfma else {}))
            self._fused_block(pk, v2p, "cross_attn", vt, Mv, att[Mp:], [self._next(pk, vs + ".attn.qkv", qkv[Mp:], 3 * inner)], ws=ws)
            # virtual self attention
            space_attn(qkv[Mp:], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, att[Mp:], inner, S, nv, nv, H,
                       dh)
            nx = [self._next(pk, p2v + ".cross_attn.to_kv", qkv[Mp:, inner:], 3 * inner, pk[p2v + ".norm_context"], 1e-5)]
            if not last:  # the virtual rows are final for this layer: project the next layer's time q|k|v right here
                nx.append(self._next(pk, nxt_qkv, qkv_nx[Mp:], 3 * inner))
            self._fused_block(pk, vs, "attn", vt, Mv, att[Mp:], nx, ws=ws)
            # point <- virtual
            space_attn(qp, inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, att[:Mp], inner, S, n, nv, H, dh)
            self._fused_block(pk, p2v, "cross_attn", pt, Mp, att[:Mp], [] if last else [self._next(pk, nxt_qkv, qkv_nx[:Mp], 3 * inner)])
            qkv, qkv_nx = qkv_nx, qkv
        od = self.out_dim
        ldh = _round_up(od, 4)
        h1, h2 = self._flow_scratch(Mp, ldh, dev)
        self._lin(pk, u + "flow_head.0", pt, h, Mp, h1, ldh, hip.ACT_RELU)
        self._lin(pk, u + "flow_head.2", h1, ldh, Mp, h2, ldh, hip.ACT_RELU)
        self._lin(pk, u + "flow_head.4", h2, ldh, Mp, delta, ldd)

    @hip.guarded
    def update_former(self, x):
        """EfficientUpdateFormer.forward on tokens x (1,N,S,D) -> (1,N,S,3+C) (test / parity entry)."""
        _, n, S, D = x.shape
        assert S == self.S and D == self.updateformer_input_dim
        pk = self._pack(x.device)
        ldx = _round_up(D, 4)
        xp = torch.zeros(n * S, ldx, device=x.device)
        xp[:, :D] = x.reshape(n * S, D)
        ldd = _round_up(self.out_dim, 4)
        delta = torch.zeros(n * S, ldd, device=x.device)
        self._update_former(pk, xp, ldx, n, delta, ldd)
        return delta[:, :self.out_dim].reshape(1, n, S, self.out_dim)

    # ------------------------------------------------------------------ one window (mvtracker.py:244-410)
    @hip.guarded
    def refine_window(self, store, frame0, coords, vis_init, track_mask, feat_init, iters=4, nan_flag=None, trace=None):
        """Iterative refinement of one window (test / parity entry; ``forward`` prepares the same state with one kernel).

        coords (n,S,3) world xyz, vis_init (n,S) logits, track_mask (n,S) {0,1}, feat_init (n,S,C).  Window slot s reads
        frame min(frame0+s, T-1).  Returns (list of coords per iteration, vis logits (n,S))."""
        coords = coords.contiguous().clone()
        ffeats = feat_init.contiguous().clone()
        mask_vis = torch.stack([track_mask.float(), vis_init.float()], dim=2).contiguous()
        return self._refine(store, frame0, coords, ffeats, mask_vis, iters, nan_flag, trace)

    def _refine(self, store, frame0, coords, ffeats, mask_vis, iters=4, nan_flag=None, trace=None, carry=None, pre_idx=None):
        """The refinement loop (mvtracker.py:350-408) on prepared state: coords (n,S,3) and ffeats (n,S,C) are updated IN PLACE,
        mask_vis (n,S,2) = (track mask, initial visibility logit).  Returns ([coords per traced iteration ..., final], vis).
        ``carry`` = (neighbour indices (L,n_prev,S,K) of the previous window's last iteration, p0): the first p0 tracks continue
        from that window, so its neighbours seed (bound) this window's first exact scan.  ``self._last_idx`` holds this window's."""
        S, C, K, L, E = self.S, self.latent_dim, self.corr_neighbors, self.corr_n_levels, self.flow_embed_dim
        n = coords.shape[0]
        dev = coords.device
        pk = self._pack(dev)
        D = self.updateformer_input_dim
        T = store["T"]
        Fc = L * K * self.corr_width
        default_corr = self.corr_n_groups == 1 and self.corr_add_neighbor_offset and not self.corr_add_neighbor_xyz
        pos = torch.empty(n, D, device=dev)
        hip.pos_embed(coords, n, S, D, _round_up(D, 6), pos, pk["pos_omega"])
        fcorr = torch.empty(n, S, Fc, device=dev)
        ldx = _round_up(D, 4)
        x = torch.empty(n * S, ldx, device=dev)      # (token_assemble writes the pad columns as zeros)
        ldd = _round_up(self.out_dim, 4)
        delta = torch.empty(n * S, ldd, device=dev)  # (out_dim columns are written and read; the pad column is never read)
        dn = torch.empty(n * S, C, device=dev)
        nsegs = [self._nseg(store["P"][lvl], K) for lvl in range(L)]
        keys = [torch.empty(n * S * nsegs[lvl] * K, device=dev, dtype=torch.int64) for lvl in range(L)]
        # neighbour indices of every level: returned for tracing AND used to seed (prune) the next exact scan
        # (``pre_idx``: the same buffer with the NEW tracks' rows of the first iteration already searched -- forward runs those
        #  unseeded searches, which depend on the query points and the geometry only, on the second stream beside the encoder)
        idx = pre_idx if pre_idx is not None else torch.empty(L, n, S, K, device=dev, dtype=torch.int32)
        assert tuple(idx.shape) == (L, n, S, K)
        grid = [tuple(store["xyz"][lvl].shape[2:4]) for lvl in range(L)]  # per-view (h, w) of each level
        preds = []
        if trace is not None:  # the state every iteration's search / correlation starts from (teacher-forced parity checks)
            trace["coords_in"], trace["ffeats_in"] = coords.clone(), ffeats.clone()
        levels = [dict(xyz=store["xyz"][lvl], P=store["P"][lvl], keys=keys[lvl], nseg=nsegs[lvl], seed_idx=idx[lvl], box=store["box"][lvl],
                       grid=store["tile_grid"][lvl], idx_out=idx[lvl], gbox=store["gbox"][lvl]) for lvl in range(L)]
        for it in range(iters):
            if it > 0:
                # every level is seeded by its own previous neighbours: the four scans are independent -> one launch
                if self.knn_one_launch:
                    hip.knn_search_levels(levels, coords, n, S, frame0, 1, T, K, seed_k=K)
                else:
                    hip.knn_scan_levels(levels, coords, n, S, frame0, 1, T, K, seed_k=K)
                    hip.knn_merge_levels(levels, n, S, K)
            else:
                n0 = 0
                if carry is not None and carry[1] > 0 and self.seed_across_windows:
                    # Tracks carried over from the previous window (mvtracker.py:648-651 start them at its last estimates): that
                    # window's final neighbours -- slot s continues slot s + S/2, the last slot held -- are K distinct points of
                    # every frame's cloud (same pixel grid), i.e. an exact upper bound of the K-th distance, and a tight one; all
                    # four levels in one seeded launch instead of four coarse-to-fine scans.  Exactness is unaffected.
                    prev_idx, n0 = carry
                    # (cached on the device: building it from a Python list is a synchronous host-to-device copy -- the host,
                    #  several milliseconds ahead of the GPU, would stall here until the stream drains)
                    slot = self._slot_cache.get((S, dev))
                    if slot is None:
                        slot = self._slot_cache[(S, dev)] = torch.tensor([min(s_ + S // 2, S - 1) for s_ in range(S)], device=dev)
                    seed_t = prev_idx[:, :n0].index_select(2, slot).contiguous()
                    lv0 = [dict(lv, keys=keys[l_][:n0 * S * nsegs[l_] * K], seed_idx=seed_t[l_], idx_out=idx[l_][:n0]) for l_, lv in enumerate(levels)]
                    if self.knn_one_launch:
                        hip.knn_search_levels(lv0, coords, n0, S, frame0, 1, T, K, seed_k=K)
                    else:
                        hip.knn_scan_levels(lv0, coords, n0, S, frame0, 1, T, K, seed_k=K)
                        hip.knn_merge_levels(lv0, n0, S, K)
                # rows [n1, n) were searched ahead of time (``pre_idx``: the tracks that enter at this window, i.e. everything behind the
                # carried ones); rows [n0, n1) still need their first, unseeded search -- empty unless the carried tracks were not seeded
                # from the previous window (``seed_across_windows`` off)
                n1 = n if pre_idx is None else (carry[1] if carry is not None else 0)
                if n0 >= n1:
                    pass
                elif self.knn_one_launch and all(b is not None for b in store["box"]):
                    # new tracks: all four levels in ONE unseeded launch (every search starts from the farthest-corner bound of the
                    # nearest full tile) -- 262 us against four dependent coarse-to-fine launches of ~100 us each
                    m = n1 - n0
                    lv1 = [dict(lv, seed_idx=None, idx_out=idx[l_][n0:n1]) for l_, lv in enumerate(levels)]
                    hip.knn_search_levels(lv1, coords[n0:n1], m, S, frame0, 1, T, K, seed_k=0)
                else:  # new tracks: coarse to fine, level l+1's neighbours bound level l's first scan
                    m = n1 - n0
                    for lvl in reversed(range(L)):
                        P = store["P"][lvl]
                        seed = {}
                        if lvl + 1 < L and grid[lvl][0] >= 2 * grid[lvl + 1][0] and grid[lvl][1] >= 2 * grid[lvl + 1][1]:
                            seed = dict(seed_idx=idx[lvl + 1][n0:n1], seed_k=K,
                                        seed_dims=(grid[lvl + 1][1], grid[lvl + 1][0], grid[lvl][1], grid[lvl][0]))
                        if self.knn_one_launch:
                            hip.knn_search(store["xyz"][lvl], P, coords[n0:n1], m, S, frame0, 1, T, K, idx[lvl][n0:n1], store["box"][lvl],
                                           grid=store["tile_grid"][lvl], gbox=store["gbox"][lvl], **seed)
                            continue
                        kl = keys[lvl][n0 * S * nsegs[lvl] * K:n1 * S * nsegs[lvl] * K]
                        hip.knn_scan(store["xyz"][lvl], P, coords[n0:n1], m, S, frame0, 1, T, K, nsegs[lvl], kl, box=store["box"][lvl],
                                     grid=store["tile_grid"][lvl], **seed)
                        hip.knn_merge(kl, m, S, K, nsegs[lvl], P, idx[lvl][n0:n1])
            if default_corr:
                hip.corr_gather_dot(store["xyz"], store["fvec"], store["P"], [idx[lvl] for lvl in range(L)], C, ffeats, coords, n, S, frame0,
                                    1, T, K, fcorr, Fc, 0)
            else:  # the reference's non-default correlation layouts (grouped dots, no offsets, neighbour coordinates)
                hip.corr_gather_dot_opts(store["xyz"], store["fvec"], store["P"], [idx[lvl] for lvl in range(L)], C, ffeats, coords, n, S,
                                         frame0, 1, T, K, self.corr_n_groups, self.corr_add_neighbor_offset, self.corr_add_neighbor_xyz,
                                         fcorr, Fc, 0)
            hook = getattr(self, "_after_first_corr", None)
            if hook is not None:  # (forward: the later frame blocks' encoder starts on the second stream now)
                self._after_first_corr = None
                hook()
            if (trace is None and "updater_struct" in pk and self.fuse_head and self.fuse_input and self.fuse_tokens
                    and pk["updater_struct"].input_frag.w):
                # everything after the correlation in ONE library call: token rows assembled inside the updater's first kernel,
                # the transformer, flow head and track / feature update (no token matrix, no delta tensor in HBM)
                ws = self._workspace(hip.updateformer_workspace_bytes(n, S), dev)
                hip.updateformer_forward_tokens(pk["updater_struct"], coords, fcorr, Fc, ffeats, C, mask_vis, pos, pk["time_embed"], E, n, None, ldd,
                                                ws, coords, ffeats, nan_flag)
                continue
            hip.token_assemble(coords, fcorr, Fc, ffeats, C, mask_vis, pos, pk["time_embed"], n, S, E, x, ldx)
            # (the delta tensor itself only leaves the fused head for tracing)
            updated = self._update_former(pk, x, ldx, n, delta if trace is not None else None, ldd, coords, ffeats, nan_flag) \
                if "updater_struct" in pk and self.fuse_head else self._update_former(pk, x, ldx, n, delta, ldd)
            if trace is not None:
                trace.setdefault("knn_idx", []).append(idx.clone())
                trace.setdefault("fcorrs", []).append(fcorr.clone())
                trace.setdefault("tokens", []).append(x[:, :D].reshape(n, S, D).clone())
                trace.setdefault("delta", []).append(delta[:, :self.out_dim].reshape(n, S, -1).clone())
            if not updated:
                hip.delta_split(delta, ldd, *pk["ffeats_norm"], coords, dn, n * S, C, nan_flag)
                self._lin(pk, "ffeats_updater.0", dn, C, n * S, ffeats, C, hip.ACT_GELU_ERF, R=ffeats, ldr=C)
            if trace is not None:  # (the intermediate estimates only feed the training loss upstream)
                preds.append(coords.clone())
                trace.setdefault("coords_iters", []).append(preds[-1])
                trace.setdefault("ffeats_iters", []).append(ffeats.clone())
        if trace is None:
            preds.append(coords)
        vis = torch.empty(n, S, device=dev)
        hip.rowdot(ffeats, C, *pk["vis"], vis, n * S, C)
        if trace is not None:
            trace["ffeats"] = ffeats
        self._last_idx = idx
        return preds, vis

    # ------------------------------------------------------------------ forward (mvtracker.py:412-732)
    @torch.no_grad()
    @hip.guarded
    def forward(
            self,
            rgbs,
            depths,
            query_points,
            intrs,
            extrs,
            iters=4,
            image_features=None,
            is_train=False,
            save_debug_logs=False,
            debug_logs_path="",
            save_rerun_logs: bool = False,
            save_rerun_logs_output_rrd_path: Optional[str] = None,
            frame_store: Optional[dict] = None,
            trace: Optional[list] = None,
            **kwargs,
    ):
        if is_train:
            raise NotImplementedError("inference only: the MI355X path has no backward")
        if save_debug_logs or save_rerun_logs:
            log.warning("save_debug_logs / save_rerun_logs are host-side visualisation hooks of the reference; ignored")
        batch_size, num_views, num_frames, _, height, width = rgbs.shape
        _, num_points, _ = query_points.shape
        assert rgbs.shape == (batch_size, num_views, num_frames, 3, height, width)
        assert depths.shape == (batch_size, num_views, num_frames, 1, height, width)
        assert query_points.shape == (batch_size, num_points, 4)
        assert intrs.shape == (batch_size, num_views, num_frames, 3, 3)
        assert extrs.shape == (batch_size, num_views, num_frames, 3, 4)
        assert batch_size == 1, "Batch size > 1 is not supported yet"
        hip.require_device(rgbs)
        dev = rgbs.device
        V, T, S, C, N = num_views, num_frames, self.S, self.latent_dim, num_points
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        # (uint8 frames -- the sample files' storage type -- stay uint8: the encoder's first kernel converts them)
        rgbs = rgbs[0].to(dev).contiguous() if rgbs.dtype == torch.uint8 else f32(rgbs[0])
        depths, intrs, extrs, query_points = map(f32, (depths[0], intrs[0], extrs[0], query_points[0]))

        # the one host sync of the call: integer query frames (mvtracker.py:489, truncation toward zero).  The read-back is
        # asynchronous (pinned buffer + event) and the query-independent geometry kernels are enqueued BEHIND it before the host
        # waits: in back-to-back calls the host wakes up when the previous call's last kernel ends, and the GPU then has the
        # geometry to run while the host enqueues the first encoder block (the kernel trace showed ~0.3 ms of idle GPU there)
        qt_dev = query_points[:, 0].long()
        geometry = None
        if dev.type == "cuda":
            qt_pin = self._pinned_i64(dev, N)
            qt_pin.copy_(qt_dev, non_blocking=True)
            ev_q = torch.cuda.Event()
            ev_q.record(torch.cuda.current_stream(dev))
            if frame_store is None:
                geometry = self.store_geometry(depths, intrs, extrs)
            ev_q.synchronize()
            qt = qt_pin.numpy().copy()
        else:
            qt = qt_dev.cpu().numpy()
        order = np.argsort(qt, kind="stable")  # mvtracker.py:514 (order among equal t is unobservable)
        qt_s = qt[order]
        # (the two tiny host-to-device copies go first, while the GPU is idle anyway: from pageable memory they block the host until
        #  the stream has drained, which after the first encoder chunk would be milliseconds)
        # (through a cached PINNED staging buffer, asynchronously: from pageable memory each copy is a host-blocking staged transfer --
        #  ~60 us of idle GPU apiece in the kernel trace.  The buffer may be rewritten by the next call: its host sync above comes
        #  after these copies in stream order)
        order_d, qt_sd = self._upload_small(dev, order.astype(np.int64), qt_s.astype(np.int32))
        # (N,3) query points sorted by start frame -- enqueued HERE, ahead of the geometry: the searches issued on the second stream
        # order themselves after the geometry event only, and they read these rows)
        qxyz = query_points[order_d, 1:].contiguous()
        if geometry is not None:
            # hand-over H4: the searches issued on the second stream order themselves after this event -- it has to come AFTER the
            # gather above (the geometry itself was enqueued, and its own event recorded, before the host sync)
            geometry["geo_event"] = torch.cuda.Event()
            geometry["geo_event"].record(torch.cuda.current_stream(dev))
        state = []

        def make_state():
            """Device-side bookkeeping of the call.  Created AFTER the first encoder chunk has been enqueued (see ``presearch``):
            the GPU idles from the host sync above until the first kernels arrive, so nothing that can wait goes before them."""
            if not state:
                state.append(dict(
                    traj=torch.zeros(T, N, 3, device=dev),        # clip outputs in the caller's query order (window_store un-sorts)
                    vis_prob=torch.zeros(T, N, device=dev), vis_logit=torch.zeros(T, N, device=dev),
                    feat_init=torch.zeros(N, C, device=dev), nan_flag=torch.zeros(1, device=dev, dtype=torch.int32)))
            return state[0]

        w = int(qt_s.min())
        windows = []
        pending = []  # (first frame, event): feature chunks being encoded on the side stream
        side_chunks = []  # first frames of the blocks the side stream has not been given yet
        pre = {}  # window start -> (neighbour buffer with the new tracks' first search done, 1-NN keys per query-frame group)

        def presearch(st):
            """Everything of the call that needs the point clouds but no features: the first, UNSEEDED neighbour search of the tracks
            that enter at each window (their coordinates are the query points, mvtracker.py:505-511) and the 1-NN scans of the
            feature init (:607-645).  Issued on the second stream between the geometry and the encoder of the first window's frames,
            so these searches (~0.4 ms at C3, latency-bound gathers) run beside the convolutions instead of after them."""
            make_state()
            if dev.type != "cuda" or not self.presearch or not self.knn_one_launch or any(b is None for b in st["box"]):
                return
            K, L = self.corr_neighbors, self.corr_n_levels
            side = self._side_stream(dev)
            main_s = torch.cuda.current_stream(dev)
            # hand-over H4: point clouds / tile boxes / sorted query rows (all enqueued on the caller's stream BEFORE the geometry event)
            side.wait_stream(main_s) if "geo_event" not in st else side.wait_event(st["geo_event"])
            self._handover(dev)
            with torch.cuda.stream(side):
                ww, q0 = w, 0
                P0 = st["P"][0]
                ns = self._nseg(P0, 1)
                while ww < T - S // 2:
                    q1 = int(np.searchsorted(qt_s, ww + S, side="left"))
                    if q1 > q0:
                        m = q1 - q0
                        idxb = torch.empty(L, q1, S, K, device=dev, dtype=torch.int32)
                        c0 = qxyz[q0:q1, None, :].expand(m, S, 3).contiguous()
                        lv = [dict(xyz=st["xyz"][l], P=st["P"][l], seed_idx=None, box=st["box"][l], grid=st["tile_grid"][l],
                                   idx_out=idxb[l][q0:], gbox=st["gbox"][l]) for l in range(L)]
                        hip.knn_search_levels(lv, c0, m, S, ww, 1, T, K, seed_k=0)
                        groups = []
                        a = q0
                        while a < q1:
                            t = int(qt_s[a])
                            b = min(int(np.searchsorted(qt_s, t, side="right")), q1)
                            keys = torch.empty((b - a) * ns, device=dev, dtype=torch.int64)
                            hip.knn_scan(st["xyz"][0], P0, qxyz[a:b], b - a, 1, t, 0, T, 1, ns, keys, box=st["box"][0], grid=st["tile_grid"][0])
                            groups.append((a, b, t, keys))
                            keys.record_stream(main_s)  # (allocated under the second stream, consumed on the caller's)
                            a = b
                        idxb.record_stream(main_s)
                        pre[ww] = (idxb, groups, c0)
                    ww += S // 2
                    q0 = q1
                ev = torch.cuda.Event()
                ev.record(side)
            pre["event"] = ev

        if w < T - S // 2:
            if frame_store is not None:
                store = frame_store
                pending = list(frame_store.get("pending", ()))  # (first frame, event) of feature blocks still in flight
            elif not self.overlap_encoder or max(w, 0) + S >= T or dev.type != "cuda":
                store = self.build_frame_store(rgbs, depths, intrs, extrs, t0=max(w, 0), after_geometry=presearch, geometry=geometry)
            else:
                # The first window needs frames [w, w+S).  The remaining frames are encoded on a second HIP stream while
                # the updater of the earlier windows runs: its kernels over the 64 virtual tracks fill a fraction of the
                # CUs, the encoder's convolutions take the rest.
                ready = max(w, 0) + S
                store = self.build_frame_store(rgbs, depths, intrs, extrs, t0=max(w, 0), t1=ready, after_geometry=presearch, geometry=geometry)
                side_chunks = list(range(ready, T, S // 2))  # first frames of the S/2-frame blocks still to encode
                if not self.defer_encoder:
                    # started by the first window BEHIND its first correlation launch (``_refine`` calls the hook): the first kernels
                    # of that window -- feature init, window state, the first correlation gather -- are latency-bound and otherwise
                    # start in the same instant as the second stream's first convolutions
                    if self.side_after_corr:
                        chunks_now = side_chunks
                        self._after_first_corr = lambda: self._encode_on_side_stream(store, rgbs, chunks_now, pending)
                    else:
                        self._encode_on_side_stream(store, rgbs, side_chunks, pending)
                    side_chunks = []
        sd_ = make_state()
        traj, vis_prob, vis_logit = sd_["traj"], sd_["vis_prob"], sd_["vis_logit"]
        feat_init, nan_flag = sd_["feat_init"], sd_["nan_flag"]
        p0 = 0
        coords = vis = prev_idx = None
        while w < T - S // 2:  # mvtracker.py:537
            p1 = int(np.searchsorted(qt_s, w + S, side="left"))  # number of queries with t < w+S (:538-540)
            assert p1 > 0
            while pending and pending[0][0] < w + S:  # the frames this window reads must have left the encoder
                torch.cuda.current_stream(dev).wait_event(pending.pop(0)[1])  # hand-over H5: feature rows encoded on the second stream
                self._handover(dev)
            if side_chunks:
                # one block of later frames per window: block j is what window j + 1 will read, so it is encoded WHILE window j is
                # refined -- its convolutions fill the CUs the 64 virtual tracks' kernels leave idle -- instead of all blocks
                # piling onto the first window (which then runs at half speed while the last windows run alone)
                self._encode_on_side_stream(store, rgbs, [side_chunks.pop(0)], pending)
            if "event" in pre:  # the searches issued ahead of time on the second stream
                torch.cuda.current_stream(dev).wait_event(pre.pop("event"))  # hand-over H6: neighbour buffers / 1-NN keys (record_stream'd)
                self._handover(dev)
            pre_w = pre.get(w)
            if p1 > p0 and pre_w is not None:  # feature init from the 1-NN keys scanned ahead of time
                P0 = store["P"][0]
                ns = self._nseg(P0, 1)
                for (a, b, t, keys) in pre_w[1]:
                    hip.knn1_gather(store["fvec"][0], P0, C, keys, b - a, ns, t, feat_init[a:b])
            elif p1 > p0:  # feature init: 1-NN in the fused level-0 cloud of the query frame (:607-645)
                P0 = store["P"][0]
                ns = self._nseg(P0, 1)
                a = p0
                while a < p1:
                    t = int(qt_s[a])
                    b = int(np.searchsorted(qt_s, t, side="right"))
                    b = min(b, p1)
                    keys = torch.empty((b - a) * ns, device=dev, dtype=torch.int64)
                    hip.knn_scan(store["xyz"][0], P0, qxyz[a:b], b - a, 1, t, 0, T, 1, ns, keys, box=store["box"][0],
                                 grid=store["tile_grid"][0])
                    hip.knn1_gather(store["fvec"][0], P0, C, keys, b - a, ns, t, feat_init[a:b])
                    a = b
            # window state in one launch: carry-over of coords / visibility LOGITS from the previous window (:648-655), track mask
            # (:505-507, :695; the repeat-last-frame padding of :598-604 is a clamped frame index), features repeated over S (:645)
            wc = torch.empty(p1, S, 3, device=dev)
            wf = torch.empty(p1, S, C, device=dev)
            wm = torch.empty(p1, S, 2, device=dev)
            hip.window_prepare(qxyz, qt_sd, feat_init, coords, vis, p1, p0, S, C, w, T, wc, wm, wf)
            wtrace = None
            if trace is not None:
                wtrace = {}
                trace.append(wtrace)
            preds, vis = self._refine(store, w, wc, wf, wm, iters=iters, nan_flag=nan_flag, trace=wtrace,
                                      carry=(prev_idx, p0) if p0 > 0 else None, pre_idx=pre_w[0] if pre_w is not None else None)
            prev_idx = self._last_idx
            coords = preds[-1]
            hip.window_store(coords, vis, order_d, p1, S, w, T, N, traj, vis_logit, vis_prob)  # :692-693, un-sorted (:710-711)
            windows.append((w, p1))
            w += S // 2
            p0 = p1
        hook = getattr(self, "_after_first_corr", None)
        if hook is not None:  # (no window ran: cannot happen while w < T - S/2, kept for safety)
            self._after_first_corr = None
            hook()
        for _, ev in pending:  # frames no window consumed: still join the side stream before the inputs are released
            torch.cuda.current_stream(dev).wait_event(ev)  # hand-over H7: the caller's inputs (read by the second stream) are released
        self.last_windows = windows
        self.last_vis_logits = vis_logit[None]
        self.last_nan_flag = nan_flag
        results = {
            "traj_e": traj[None],
            "feat_init": feat_init[None, None].expand(1, S, -1, -1),
            "vis_e": vis_prob[None],
        }
        return results

    def check_finite(self):
        """Deferred NaN guard (reference mvtracker.py:401-404): raises if the last forward produced NaN tracks."""
        if int(self.last_nan_flag.item()) != 0:
            raise FloatingPointError("Got NaN values in coords, perhaps the training exploded")
