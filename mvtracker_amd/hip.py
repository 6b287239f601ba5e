"""ctypes binding of libmvtracker_hip.so (C ABI in include/mvtracker_hip.h).

The library is the product: there is NO eager / CPU fallback.  Importing this module loads the
shared object and raises ``RuntimeError`` if it is missing (build it with
``python -m mvtracker_amd.build``); every wrapper raises on a non-zero return code.  PyTorch only
provides device memory and the current HIP stream.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVT_LIB") or os.path.join(_HERE, "lib", "libmvtracker_hip.so")  # MVT_LIB: A/B builds of the same ABI

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f"{LIB_PATH} not found: the HIP library is required (no fallback path exists). "
        "Build it with `python -m mvtracker_amd.build`.")
_lib = C.CDLL(LIB_PATH)

P, I, LL, F = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> argument types (all return int unless listed in _RET)
SIGNATURES = {
    "mvt_abi_version": [],
    "mvt_build_arch": [],
    "mvt_gemm": [P, I, P, I, P, P, I, P, I, I, I, I, I, P],
    "mvt_conv2d": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "mvt_split_bf16": [P, P, P, LL, P],
    "mvt_gemm_bf16": [P, I, P, P, I, P, P, I, P, I, I, I, I, I, I, P],
    "mvt_conv2d_stat_slots": [I, I, I, I, I, I, I, I],
    "mvt_conv2d_bf16": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, P],
    "mvt_conv3x3s2_down_bf16": [P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, P],
    "mvt_instnorm_finish_slots": [P, I, P, I, LL, I, P],
    "mvt_ln_gemm_bf16": [P, I, P, P, F, P, P, I, P, P, I, P, I, I, I, I, I, P],
    "mvt_pack_frag_bf16": [P, I, I, I, P, P],
    "mvt_block_fused_bf16": [P, I, P, I, I, I, P, I, P, P, I, P, P, I, P, I, P, I, LL, I, P, P],
    "mvt_mlp_fused_bf16": [P, I, P, I, P, P, I, P, LL, I, I, F, P],
    "mvt_rgb_to_nhwc4": [P, P, I, I, I, I, I, I, P],
    "mvt_rgb_u8_to_nhwc4": [P, P, I, I, I, I, I, I, P],
    "mvt_rgb_images_to_nhwc4": [P, I, P, I, I, I, I, LL, I, P],
    "mvt_resize_nearest": [P, P, LL, I, I, I, I, P],
    "mvt_instnorm_stats": [P, I, P, P, I, LL, I, I, P],
    "mvt_instnorm_apply": [P, P, P, P, P, I, LL, I, I, P],
    "mvt_resize_bilinear_ac": [P, P, I, I, I, I, I, I, I, I, I, P],
    "mvt_concat_resize_bilinear_ac": [I, P, P, P, P, P, I, I, I, I, I, P],
    "mvt_invert_cameras": [P, P, P, P, I, P],
    "mvt_depth_subsample": [P, P, I, I, I, I, I, P],
    "mvt_avgpool2": [P, P, LL, I, I, I, I, P],
    "mvt_unproject": [P, P, P, P, I, I, I, I, I, I, P],
    "mvt_tile_aabb": [P, LL, I, I, I, P, P],
    "mvt_knn_scan": [P, LL, P, I, I, I, I, I, I, I, P, P, I, I, I, I, I, P, I, I, P],
    "mvt_knn_merge": [P, I, I, I, I, LL, P, P],
    "mvt_ln_proj_bf16": [P, I, P, I, LL, I, P],
    "mvt_input_proj_bf16": [P, I, I, LL, P, P, P, I, P, I, P, I, LL, I, P],
    "mvt_adapter_best_view": [P, P, P, P, I, I, I, I, I, P, P, P],
    "mvt_knn_scan_levels": [I, P, P, I, I, I, I, I, I, I, P],
    "mvt_knn_search_levels": [I, P, P, I, I, I, I, I, I, I, P],
    "mvt_knn_search": [P, LL, P, I, I, I, I, I, I, P, I, I, I, I, I, P, P, I, I, P, P],
    "mvt_tile_group_aabb": [P, LL, I, P, P],
    "mvt_knn_merge_levels": [I, P, I, I, I, P],
    "mvt_corr_gather_dot": [I, P, P, I, P, P, I, P, P, I, I, I, I, I, I, P, I, I, P],
    "mvt_corr_gather_dot_opts": [I, P, P, I, P, P, I, P, P, I, I, I, I, I, I, I, I, I, P, I, I, P],
    "mvt_knn1_gather": [P, I, LL, I, P, I, I, I, P, P, P],
    "mvt_window_corr": [P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "mvt_window_corr_levels": [I, P, I, P, P, P, P, P, I, I, I, I, I, I, P],
    "mvt_pos_embed": [P, I, I, I, I, P, P, P],
    "mvt_token_assemble": [P, P, I, P, I, P, P, P, I, I, I, P, I, P],
    "mvt_delta_split": [P, I, P, P, P, P, LL, I, P, P],
    "mvt_rowdot": [P, I, P, P, P, LL, I, P],
    "mvt_layernorm": [P, I, P, P, P, I, LL, I, F, P],
    "mvt_attention": [P, I, LL, LL, P, P, I, LL, LL, P, I, I, I, I, I, I, P],
    "mvt_attention_bf16": [P, I, LL, LL, P, P, I, LL, LL, P, I, I, I, I, I, I, I, P, P],
    "mvt_broadcast_rows": [P, P, I, I, I, I, P],
    "mvt_attn_block_fused_bf16": [P, I, P, P, P, P, P, P, P, I, P, I, LL, I, P, P],
    "mvt_window_prepare": [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P],
    "mvt_window_store": [P, P, P, I, I, I, I, I, P, P, P, P],
    "mvt_track_metrics": [P, P, P, P, P, I, I, I, P, I, F, P, I, P],
    "mvt_encoder_workspace_bytes": [I, I, I, I],
    "mvt_encoder_forward": [P, P, I, I, I, P, I, I, P, LL, P],
    "mvt_encoder_forward_rgb": [P, P, I, I, I, LL, I, I, I, P, I, I, P, LL, P],
    "mvt_updateformer_workspace_bytes": [I, I],
    "mvt_updateformer_forward": [P, P, I, I, P, I, P, P, P, P, LL, P],
    "mvt_updateformer_forward_tokens": [P, P, I, P, I, P, P, P, P, LL, P],
    "mvt_token_input_proj_bf16": [P, P, I, P, I, P, P, P, I, I, I, P, P, P, P, I, P, I, LL, I, P],
    "mvt_update_head_bf16": [P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, LL, I, I, P, P],
}
_RET = {"mvt_build_arch": C.c_char_p, "mvt_encoder_workspace_bytes": C.c_longlong, "mvt_updateformer_workspace_bytes": C.c_longlong}

for _name, _args in SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here = header / library mismatch
    _fn.argtypes = _args
    _fn.restype = _RET.get(_name, C.c_int)

ACT_NONE, ACT_RELU, ACT_GELU_TANH, ACT_GELU_ERF = 0, 1, 2, 3
IN_SLABS = 64


class HipError(RuntimeError):
    pass


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("libmvtracker_hip needs device tensors (got a CPU tensor); there is no CPU path")
    return t.data_ptr()


def require_device(t):
    """The product path has no CPU implementation: refuse host tensors loudly; and the tensors' device must be the current
    one (the launch stream is the current device's) -- entry points get there through ``device_guard``."""
    if not t.is_cuda:
        raise HipError("the MI355X tracker needs device tensors (got a CPU tensor); there is no CPU path")
    if t.device.index != torch.cuda.current_device():
        raise HipError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                       "launches go to the current device's stream (wrap the call in hip.device_guard(tensor))")


def _stream():
    """The launch stream: torch's current stream of the CURRENT device.  Every public entry point of the package runs
    under ``device_guard`` of its tensors, so this is the stream of the device the pointers live on."""
    return torch.cuda.current_stream().cuda_stream


def device_guard(t):
    """Context manager that makes ``t``'s device the current one (no-op for host tensors, which only the CPU host-logic
    tests pass).  A model on cuda:1 called while cuda:0 is current would otherwise launch on device 0's stream with
    device-1 pointers."""
    if t is not None and t.is_cuda:
        return torch.cuda.device(t.device)
    return contextlib.nullcontext()


def guarded(fn):
    """Decorator for entry points whose first tensor argument fixes the device of the call."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t = next((a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)), None)
        if t is None:  # e.g. (store dict, ...) first: look one level into dicts / lists
            for a in list(args) + list(kwargs.values()):
                if isinstance(a, dict):
                    a = list(a.values())
                if isinstance(a, (list, tuple)):
                    t = next((x for x in a if isinstance(x, torch.Tensor)), None) or next(
                        (y for x in a if isinstance(x, (list, tuple)) for y in x if isinstance(y, torch.Tensor)), None)
                    if t is not None:
                        break
        with device_guard(t):
            return fn(*args, **kwargs)
    return wrapper


def _call(name, *args):
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise HipError(f"{name} failed with code {rc}" + (" (arguments rejected)" if rc == 1 else " (HIP launch error)"))


IO_IN_BF16, IO_OUT_BF16, IO_SHORT_WG = 1, 2, 4


def _io(t_in, t_out=None):
    """MVT_IO_* flags from the element types of the activation tensors (fp32 or bf16)."""
    for t in (t_in, t_out):
        assert t is None or t.dtype in (torch.float32, torch.bfloat16), t.dtype
    return (IO_IN_BF16 if t_in is not None and t_in.dtype == torch.bfloat16 else 0) | \
           (IO_OUT_BF16 if t_out is not None and t_out.dtype == torch.bfloat16 else 0)


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def abi_version() -> int:
    return _lib.mvt_abi_version()


def build_arch() -> str:
    return _lib.mvt_build_arch().decode()


# ------------------------------------------------------------------ thin typed wrappers
def gemm(A, lda, Wt, ldw, bias, R, ldr, Cm, ldc, M, N, K, act=ACT_NONE):
    _call("mvt_gemm", _ptr(A), lda, _ptr(Wt), ldw, _ptr(bias), _ptr(R), ldr, _ptr(Cm), ldc, M, N, K, act, _stream())


def conv2d(x, wt, bias, out, n, H, W, Cin, Cout, KH, KW, stride, pad, ldo, act=ACT_NONE):
    _call("mvt_conv2d", _ptr(x), _ptr(wt), _ptr(bias), _ptr(out), n, H, W, Cin, Cout, KH, KW, stride, pad, ldo, act, _stream())


def split_bf16(src, hi, lo, n):
    _call("mvt_split_bf16", _ptr(src), _ptr(hi), _ptr(lo), n, _stream())


def gemm_bf16(A, lda, Whi, Wlo, ldw, bias, R, ldr, Cm, ldc, M, N, K, act=ACT_NONE):
    _call("mvt_gemm_bf16", _ptr(A), lda, _ptr(Whi), _ptr(Wlo), ldw, _ptr(bias), _ptr(R), ldr, _ptr(Cm), ldc, M, N, K, act,
          _io(None, Cm), _stream())


def conv2d_stat_slots(H, W, Cin, KH, KW, stride, pad, split=False) -> int:
    """split: the weights carry a bf16 lo part (bf16x3 mode) -- the two 3x3 kernels cut the image differently."""
    return _lib.mvt_conv2d_stat_slots(H, W, Cin, KH, KW, stride, pad, int(split))


def conv2d_bf16(x, wt_hi, wt_lo, bias, out, n, H, W, Cin, Cout, KH, KW, stride, pad, ldo, act=ACT_NONE, in_stats=None,
                out_partial=None, short_wg=False):
    """``short_wg`` (MVT_IO_SHORT_WG): the launch shares the GPU with other streams -- keep every workgroup short-lived (the wide
    3x3 layers then run on the 64-channel row tiles instead of one 512-thread workgroup per CU; identical results)."""
    _call("mvt_conv2d_bf16", _ptr(x), _ptr(wt_hi), _ptr(wt_lo), _ptr(bias), _ptr(out), n, H, W, Cin, Cout, KH, KW, stride, pad,
          ldo, act, _io(x, out) | (IO_SHORT_WG if short_wg else 0), _ptr(in_stats), _ptr(out_partial), _stream())


def conv3x3s2_down_bf16(x, w3, b3, wd, bd, out3, outd, n, H, W, Cin, Cout, ldo, part3=None, partd=None):
    """conv1 (3x3 / stride 2) and downsample[0] (1x1 / stride 2) of a strided ResidualBlock in one launch (bf16 tensors)."""
    assert x.dtype == torch.bfloat16 and out3.dtype == torch.bfloat16 and outd.dtype == torch.bfloat16
    _call("mvt_conv3x3s2_down_bf16", _ptr(x), _ptr(w3), _ptr(b3), _ptr(wd), _ptr(bd), _ptr(out3), _ptr(outd), n, H, W, Cin, Cout, ldo,
          _ptr(part3), _ptr(partd), _stream())


def instnorm_finish_slots(partial, slots, mean_rstd, n, HW, Cc):
    _call("mvt_instnorm_finish_slots", _ptr(partial), slots, _ptr(mean_rstd), n, HW, Cc, _stream())


def ln_gemm_bf16(A, lda, ln_w, ln_b, eps, Whi, Wlo, ldw, bias, R, ldr, Cm, ldc, M, N, K, act=ACT_NONE):
    _call("mvt_ln_gemm_bf16", _ptr(A), lda, _ptr(ln_w), _ptr(ln_b), eps, _ptr(Whi), _ptr(Wlo), ldw, _ptr(bias), _ptr(R), ldr,
          _ptr(Cm), ldc, M, N, K, act, _stream())


def pack_frag_bf16(w, ld, N, K, out):
    _call("mvt_pack_frag_bf16", _ptr(w), ld, N, K, _ptr(out), _stream())


class BlockNext(C.Structure):
    """mvt_block_next of include/mvtracker_hip.h."""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("lnw", C.c_void_p), ("lnb", C.c_void_p), ("y", C.c_void_p),
                ("ldw", C.c_int), ("N", C.c_int), ("ldy", C.c_int), ("eps", C.c_float), ("row_lo", C.c_longlong),
                ("row_hi", C.c_longlong), ("y_bf16", C.c_int)]


BLOCK_MAX_NEXT = 3


def block_fused_bf16(x, ldx, att, ldatt, Ko, wo, ldwo, bo, w1, ldw1, b1, w2, ldw2, b2, H, nexts, M, Cc, ws=None):
    """nexts: list of dicts(w, ldw, b, N, lnw, lnb, eps, y, ldy[, rows=(lo, hi)]) -- at most BLOCK_MAX_NEXT follow-up
    projections; ``rows`` restricts one to a row range (y is always indexed by the global row).  ``ws``: optional fp32
    workspace of (H/256 + 1) * M * Cc elements that lets M <= 2048 run cut over 4x more workgroups."""
    assert ws is None or (ws.dtype == torch.float32 and ws.numel() >= (H // 256 + 1) * M * Cc)
    arr = (BlockNext * max(1, len(nexts)))()
    for i, nx in enumerate(nexts):
        arr[i] = BlockNext(_ptr(nx["w"]), _ptr(nx["b"]), _ptr(nx.get("lnw")), _ptr(nx.get("lnb")), _ptr(nx["y"]), nx["ldw"], nx["N"],
                           nx["ldy"], nx["eps"], *nx.get("rows", (0, 0)), 1 if nx["y"].dtype == torch.bfloat16 else 0)
    _call("mvt_block_fused_bf16", _ptr(x), ldx, _ptr(att), 1 if (att is not None and att.dtype == torch.bfloat16) else 0, ldatt, Ko, _ptr(wo), ldwo, _ptr(bo), _ptr(w1), ldw1, _ptr(b1), _ptr(w2),
          ldw2, _ptr(b2), H, C.cast(arr, C.c_void_p), len(nexts), M, Cc, _ptr(ws), _stream())


def ln_proj_bf16(x, ldx, nexts, M, Cc):
    """LayerNorm + projections of ``nexts`` (same dicts as ``block_fused_bf16``); x is only read."""
    arr = (BlockNext * len(nexts))()
    for i, nx in enumerate(nexts):
        arr[i] = BlockNext(_ptr(nx["w"]), _ptr(nx["b"]), _ptr(nx.get("lnw")), _ptr(nx.get("lnb")), _ptr(nx["y"]), nx["ldw"], nx["N"],
                           nx["ldy"], nx["eps"], *nx.get("rows", (0, 0)), 1 if nx["y"].dtype == torch.bfloat16 else 0)
    _call("mvt_ln_proj_bf16", _ptr(x), ldx, C.cast(arr, C.c_void_p), len(nexts), M, Cc, _stream())


def input_proj_bf16(tokens, ldtok, token_dim, Mp, win, bin_, virtual_tokens, S, x, ldx, nexts, M, Cc):
    """x = [tokens . Win^T + b ; virtual tokens] written, then the LayerNorm + projections of ``nexts`` (one launch)."""
    arr = (BlockNext * len(nexts))()
    for i, nx in enumerate(nexts):
        arr[i] = BlockNext(_ptr(nx["w"]), _ptr(nx["b"]), _ptr(nx.get("lnw")), _ptr(nx.get("lnb")), _ptr(nx["y"]), nx["ldw"], nx["N"],
                           nx["ldy"], nx["eps"], *nx.get("rows", (0, 0)), 1 if nx["y"].dtype == torch.bfloat16 else 0)
    _call("mvt_input_proj_bf16", _ptr(_f32c(tokens)), ldtok, token_dim, Mp, _ptr(win), _ptr(bin_), _ptr(virtual_tokens), S, _ptr(x), ldx,
          C.cast(arr, C.c_void_p), len(nexts), M, Cc, _stream())


def mlp_fused_bf16(x, ldx, w1, ldw1, b1, w2, ldw2, b2, M, Cc, H, eps):
    _call("mvt_mlp_fused_bf16", _ptr(x), ldx, _ptr(w1), ldw1, _ptr(b1), _ptr(w2), ldw2, _ptr(b2), M, Cc, H, eps, _stream())


def rgb_to_nhwc4(rgbs, out, V, T, H, W, t0, nt):
    if rgbs.dtype == torch.uint8:
        assert rgbs.is_contiguous()
        _call("mvt_rgb_u8_to_nhwc4", _ptr(rgbs), _ptr(out), V, T, H, W, t0, nt, _stream())
    else:
        _call("mvt_rgb_to_nhwc4", _ptr(_f32c(rgbs)), _ptr(out), V, T, H, W, t0, nt, _stream())


def rgb_images_to_nhwc4(rgbs, out, V, T, H, W, img0, nimg):
    """Images img0 .. img0+nimg-1 in frame-major numbering (t * V + v) of rgbs (V,T,3,H,W) fp32 or uint8 -> out (nimg,H,W,4)."""
    assert rgbs.is_contiguous() and rgbs.dtype in (torch.uint8, torch.float32)
    _call("mvt_rgb_images_to_nhwc4", _ptr(rgbs), 1 if rgbs.dtype == torch.uint8 else 0, _ptr(out), V, T, H, W, img0, nimg, _stream())


def resize_nearest(x, out, planes, Hi, Wi, Ho, Wo):
    _call("mvt_resize_nearest", _ptr(_f32c(x)), _ptr(out), planes, Hi, Wi, Ho, Wo, _stream())


def instnorm_stats(x, ldx, partial, mean_rstd, n, HW, Cc):
    _call("mvt_instnorm_stats", _ptr(x), ldx, _ptr(partial), _ptr(mean_rstd), n, HW, Cc, _io(x), _stream())


def instnorm_apply(x, mean_rstd, skip, skip_stats, y, n, HW, Cc, skip_relu=False):
    assert y.dtype == x.dtype and (skip is None or skip.dtype == x.dtype)
    _call("mvt_instnorm_apply", _ptr(x), _ptr(mean_rstd), _ptr(skip), _ptr(skip_stats), _ptr(y), n, HW, Cc,
          _io(x, y) | (4 if skip_relu else 0), _stream())


def resize_bilinear_ac(src, dst, n, Hs, Ws, Cc, Hd, Wd, ldd, c_off):
    assert src.dtype == dst.dtype
    _call("mvt_resize_bilinear_ac", _ptr(src), _ptr(dst), n, Hs, Ws, Cc, Hd, Wd, ldd, c_off, _io(src, dst), _stream())


def concat_resize_bilinear_ac(srcs, dims, dst, n, Hd, Wd, ldd):
    """srcs: device tensors (n, Hs_k, Ws_k, C_k) of one element type; dims: [(Hs_k, Ws_k, C_k)]; one launch for the whole concat."""
    k = len(srcs)
    assert all(t.dtype == dst.dtype for t in srcs)
    ia = lambda j: (C.c_int * k)(*[d[j] for d in dims])
    _call("mvt_concat_resize_bilinear_ac", k, (C.c_void_p * k)(*[_ptr(t) for t in srcs]), ia(0), ia(1), ia(2), _ptr(dst), n, Hd, Wd, ldd,
          _io(srcs[0], dst), _stream())


def invert_cameras(intrs, extrs, kinv, einv, n):
    _call("mvt_invert_cameras", _ptr(_f32c(intrs)), _ptr(_f32c(extrs)), _ptr(kinv), _ptr(einv), n, _stream())


def depth_subsample(depths, out, V, T, H, W, s):
    _call("mvt_depth_subsample", _ptr(_f32c(depths)), _ptr(out), V, T, H, W, s, _stream())


def avgpool2(x, out, n, h, w, Cc):
    assert x.dtype == out.dtype
    _call("mvt_avgpool2", _ptr(x), _ptr(out), n, h, w, Cc, _io(x, out), _stream())


def unproject(depth_s, kinv, einv, xyz, V, T, hs, ws, stride, level):
    _call("mvt_unproject", _ptr(depth_s), _ptr(kinv), _ptr(einv), _ptr(xyz), V, T, hs, ws, stride, level, _stream())


def tile_aabb(xyz, Pn, T, box, grid=(0, 0)):
    """box (T, ceil(Pn/64), 8) <- bounding boxes of the 64-point tiles; grid = per-view (w, h) for 8x8 patch tiles."""
    _call("mvt_tile_aabb", _ptr(xyz), Pn, T, grid[0], grid[1], _ptr(box), _stream())


def knn_scan(xyz, Pn, coords, N, S, frame0, frame_step, T, K, nseg, keys, seed_idx=None, seed_k=0, seed_dims=(0, 0, 0, 0), box=None,
             grid=(0, 0)):
    """seed_dims = (coarse_w, coarse_h, fine_w, fine_h) per-view grids when the seed comes from the coarser level;
    box / grid: tile bounding boxes from ``tile_aabb`` (same grid) for culling."""
    _call("mvt_knn_scan", _ptr(xyz), Pn, _ptr(coords), N, S, frame0, frame_step, T, K, nseg, _ptr(keys), _ptr(seed_idx), seed_k,
          *seed_dims, _ptr(box), grid[0], grid[1], _stream())


class KnnLevel(C.Structure):
    """mvt_knn_level of include/mvtracker_hip.h."""
    _fields_ = [("xyz", C.c_void_p), ("P", C.c_longlong), ("keys", C.c_void_p), ("seed_idx", C.c_void_p), ("tile_box", C.c_void_p),
                ("nseg", C.c_int), ("grid_w", C.c_int), ("grid_h", C.c_int), ("idx_out", C.c_void_p), ("group_box", C.c_void_p)]


def _knn_levels(levels):
    arr = (KnnLevel * len(levels))()
    for i, lv in enumerate(levels):
        g = lv.get("grid", (0, 0))
        arr[i] = KnnLevel(_ptr(lv["xyz"]), lv["P"], _ptr(lv["keys"]), _ptr(lv.get("seed_idx")), _ptr(lv.get("box")), lv["nseg"], g[0], g[1],
                          _ptr(lv.get("idx_out")), _ptr(lv.get("gbox")))
    return arr


def knn_scan_levels(levels, coords, N, S, frame0, frame_step, T, K, seed_k=0):
    """levels: list of dicts(xyz, P, keys, nseg[, seed_idx, box, grid, idx_out]); one launch for all of them."""
    arr = _knn_levels(levels)
    _call("mvt_knn_scan_levels", len(levels), C.cast(arr, C.c_void_p), _ptr(coords), N, S, frame0, frame_step, T, K, seed_k, _stream())


def tile_group_aabb(box, Pn, T, group_box):
    """group_box [T][ceil(ntiles/64)][8]: union boxes of 64 consecutive tiles (coarse culling level of the searches)."""
    _call("mvt_tile_group_aabb", _ptr(box), Pn, T, _ptr(group_box), _stream())


def knn_search(xyz, Pn, coords, N, S, frame0, frame_step, T, K, idx_out, box, grid=(0, 0), gbox=None, seed_idx=None, seed_k=0,
               seed_dims=(0, 0, 0, 0)):
    """``knn_scan`` (one segment) + ``knn_merge`` in one launch: neighbour indices straight to idx_out (N,S,K) int32."""
    _call("mvt_knn_search", _ptr(xyz), Pn, _ptr(coords), N, S, frame0, frame_step, T, K, _ptr(seed_idx), seed_k, *seed_dims, _ptr(box),
          _ptr(gbox), grid[0], grid[1], _ptr(idx_out), _stream())


def knn_search_levels(levels, coords, N, S, frame0, frame_step, T, K, seed_k):
    """Seeded scan + merge in one launch: levels of dicts(xyz, P, seed_idx, box, grid, idx_out[, gbox]); idx_out may alias seed_idx."""
    arr = _knn_levels([dict(lv, keys=None, nseg=1) for lv in levels])
    _call("mvt_knn_search_levels", len(levels), C.cast(arr, C.c_void_p), _ptr(coords), N, S, frame0, frame_step, T, K, seed_k, _stream())


def knn_merge_levels(levels, N, S, K):
    arr = _knn_levels(levels)
    _call("mvt_knn_merge_levels", len(levels), C.cast(arr, C.c_void_p), N, S, K, _stream())


def adapter_best_view(depths, intrs, extrs, query_points, V, T, H, W, N, view_out, xyz_out=None):
    _call("mvt_adapter_best_view", _ptr(_f32c(depths)), _ptr(_f32c(intrs)), _ptr(_f32c(extrs)), _ptr(_f32c(query_points)), V, T, H, W, N,
          _ptr(view_out), _ptr(xyz_out), _stream())


def knn_merge(keys, N, S, K, nseg, Pn, idx_out):
    _call("mvt_knn_merge", _ptr(keys), N, S, K, nseg, Pn, _ptr(idx_out), _stream())


def corr_gather_dot(xyz_l, fvec_l, P_l, idx_l, Cc, targets, coords, N, S, frame0, frame_step, T, K, out, ldo, o_off):
    """xyz_l / fvec_l / idx_l: lists of per-level device tensors; P_l: list of point counts."""
    n = len(xyz_l)
    pa = (C.c_void_p * n)
    bf = fvec_l[0].dtype == torch.bfloat16
    assert all((t.dtype == torch.bfloat16) == bf for t in fvec_l)
    _call("mvt_corr_gather_dot", n, pa(*[_ptr(t) for t in xyz_l]), pa(*[_ptr(t) for t in fvec_l]), 1 if bf else 0, (C.c_longlong * n)(*P_l),
          pa(*[_ptr(t) for t in idx_l]), Cc, _ptr(targets), _ptr(coords), N, S, frame0, frame_step, T, K, _ptr(out), ldo, o_off,
          _stream())


def corr_gather_dot_opts(xyz_l, fvec_l, P_l, idx_l, Cc, targets, coords, N, S, frame0, frame_step, T, K, groups, add_offset, add_xyz, out, ldo,
                         o_off):
    """``corr_gather_dot`` with the non-default correlation options: groups + 3 add_offset + 3 add_xyz values per neighbour."""
    n = len(xyz_l)
    pa = (C.c_void_p * n)
    bf = fvec_l[0].dtype == torch.bfloat16
    assert all((t.dtype == torch.bfloat16) == bf for t in fvec_l)
    _call("mvt_corr_gather_dot_opts", n, pa(*[_ptr(t) for t in xyz_l]), pa(*[_ptr(t) for t in fvec_l]), 1 if bf else 0, (C.c_longlong * n)(*P_l),
          pa(*[_ptr(t) for t in idx_l]), Cc, _ptr(targets), _ptr(coords), N, S, frame0, frame_step, T, K, groups, 1 if add_offset else 0,
          1 if add_xyz else 0, _ptr(out), ldo, o_off, _stream())


def knn1_gather(fvec, Pn, Cc, keys, n, nseg, frame, feat_out, idx_out=None):
    _call("mvt_knn1_gather", _ptr(fvec), 1 if fvec.dtype == torch.bfloat16 else 0, Pn, Cc, _ptr(keys), n, nseg, frame, _ptr(feat_out),
          _ptr(idx_out), _stream())


def window_corr(fmap, targets, coords, out, BS, N, Cc, h, w, level, radius, ldo, o_off):
    _call("mvt_window_corr", _ptr(fmap), _ptr(targets), _ptr(coords), _ptr(out), BS, N, Cc, h, w, level, radius, ldo, o_off,
          _stream())


def window_corr_levels(fmaps, targets, coords, out, BS, N, Cc, radius, ldo, o_off=0):
    """CorrBlock.corr_sample for all levels in one launch; fmaps: list of channels-last (BS,h,w,C) device tensors, fp32 or bf16."""
    n = len(fmaps)
    bf = fmaps[0].dtype == torch.bfloat16
    assert all((f.dtype == torch.bfloat16) == bf and f.is_contiguous() and f.shape[0] == BS and f.shape[3] == Cc for f in fmaps)
    _call("mvt_window_corr_levels", n, (C.c_void_p * n)(*[_ptr(f) for f in fmaps]), 1 if bf else 0, (C.c_int * n)(*[f.shape[1] for f in fmaps]),
          (C.c_int * n)(*[f.shape[2] for f in fmaps]), _ptr(_f32c(targets)), _ptr(_f32c(coords)), _ptr(out), BS, N, Cc, radius, ldo, o_off, _stream())


def pos_embed(coords, N, S, D, dim_padded, pos, omega=None):
    """omega: optional device fp64 table of dim_padded // 6 frequencies (the reference's numpy values)."""
    assert omega is None or (omega.dtype == torch.float64 and omega.numel() >= dim_padded // 6)
    _call("mvt_pos_embed", _ptr(coords), N, S, D, dim_padded, _ptr(omega), _ptr(pos), _stream())


def token_assemble(coords, fcorr, Fc, ffeats, Cc, mask_vis, pos, time_embed, N, S, E, x, ldx):
    _call("mvt_token_assemble", _ptr(coords), _ptr(fcorr), Fc, _ptr(ffeats), Cc, _ptr(mask_vis), _ptr(pos), _ptr(time_embed),
          N, S, E, _ptr(x), ldx, _stream())


def delta_split(delta, ldd, gw, gb, coords, dn, rows, Cc, nan_flag=None):
    _call("mvt_delta_split", _ptr(delta), ldd, _ptr(gw), _ptr(gb), _ptr(coords), _ptr(dn), rows, Cc, _ptr(nan_flag), _stream())


def rowdot(x, ldx, w, b, out, rows, Cc):
    _call("mvt_rowdot", _ptr(x), ldx, _ptr(w), _ptr(b), _ptr(out), rows, Cc, _stream())


def layernorm(x, ldx, w, b, y, ldy, rows, Cc, eps):
    _call("mvt_layernorm", _ptr(x), ldx, _ptr(w), _ptr(b), _ptr(y), ldy, rows, Cc, eps, _stream())


def attention(q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, o, ldo, groups, nq, nk, heads, dh):
    _call("mvt_attention", _ptr(q), ldq, q_gs, q_is, _ptr(k), _ptr(v), ldkv, k_gs, k_is, _ptr(o), ldo, groups, nq, nk, heads,
          dh, _stream())


def attention_ws_floats(groups, nq, heads):
    """Workspace size (fp32 elements) of the key-split path of ``attention_bf16``."""
    return 4 * groups * heads * ((nq + 63) // 64) * 64 * 68


def attention_bf16(q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, o, ldo, groups, nq, nk, heads, dh, ws=None):
    assert q.dtype == k.dtype == v.dtype
    assert ws is None or (ws.dtype == torch.float32 and ws.numel() >= attention_ws_floats(groups, nq, heads))
    _call("mvt_attention_bf16", _ptr(q), ldq, q_gs, q_is, _ptr(k), _ptr(v), ldkv, k_gs, k_is, _ptr(o), ldo, groups, nq, nk, heads,
          dh, _io(q, o), _ptr(ws), _stream())


def window_prepare(qxyz, qt, feat_init, prev_coords, prev_vis, n, p0, S, Cc, w, T, coords, mask_vis, ffeats):
    assert qt.dtype == torch.int32
    _call("mvt_window_prepare", _ptr(_f32c(qxyz)), _ptr(qt), _ptr(_f32c(feat_init)), _ptr(prev_coords), _ptr(prev_vis), n, p0, S, Cc, w, T,
          _ptr(coords), _ptr(mask_vis), _ptr(ffeats), _stream())


def window_store(coords, vis, order, n, S, w, T, N, traj, vis_logit, vis_prob):
    assert order.dtype == torch.int64
    _call("mvt_window_store", _ptr(coords), _ptr(vis), _ptr(order), n, S, w, T, N, _ptr(traj), _ptr(vis_logit), _ptr(vis_prob), _stream())


def broadcast_rows(v, x, ld, n, S, Cc):
    _call("mvt_broadcast_rows", _ptr(v), _ptr(x), ld, n, S, Cc, _stream())


# ------------------------------------------------------------------ composite entry points
class LinFrag(C.Structure):
    """mvt_lin_frag of include/mvtracker_hip.h."""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("N", C.c_int), ("K", C.c_int)]


class LinRows(C.Structure):
    """mvt_lin_rows."""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("ldw", C.c_int)]


class UpdaterBlock(C.Structure):
    """mvt_updater_block."""
    _fields_ = [("qkv", LinFrag), ("q", LinFrag), ("kv", LinFrag), ("ctx_ln_w", C.c_void_p), ("ctx_ln_b", C.c_void_p),
                ("out", LinFrag), ("fc1", LinFrag), ("fc2", LinFrag)]


UPDATER_MAX_DEPTH = 8
COMPOSITE = True  # (the CPU host-logic tests, which replace the kernels one by one, switch the composite calls off)


class UpdaterWeights(C.Structure):
    """mvt_updater_weights: host struct of device pointers (the tensors must outlive it: keep them referenced)."""
    _fields_ = [("depth", C.c_int), ("hidden", C.c_int), ("heads", C.c_int), ("dim_head", C.c_int), ("n_virtual", C.c_int),
                ("S", C.c_int), ("token_dim", C.c_int), ("out_dim", C.c_int), ("fuse_attention", C.c_int), ("virtual_tokens", C.c_void_p),
                ("input_transform", LinRows), ("flow0", LinRows), ("flow2", LinRows), ("flow4", LinRows),
                ("time_blk", UpdaterBlock * UPDATER_MAX_DEPTH), ("v2p", UpdaterBlock * UPDATER_MAX_DEPTH),
                ("vself", UpdaterBlock * UPDATER_MAX_DEPTH), ("p2v", UpdaterBlock * UPDATER_MAX_DEPTH),
                ("flow0_frag", LinFrag), ("flow2_frag", LinFrag), ("flow4_frag", LinFrag), ("ffeats_updater", LinFrag),
                ("input_frag", LinFrag),
                ("ffeats_norm_w", C.c_void_p), ("ffeats_norm_b", C.c_void_p)]


def lin_frag(frag, bias, N, K):
    return LinFrag(_ptr(frag), _ptr(bias), N, K)


def lin_rows(w_hi, bias, N, K):
    return LinRows(_ptr(w_hi), _ptr(bias), N, K, w_hi.shape[1])


def updateformer_workspace_bytes(n, S) -> int:
    return int(_lib.mvt_updateformer_workspace_bytes(n, S))


def updateformer_forward(weights: UpdaterWeights, x, ldx, n, delta, ldd, workspace, coords=None, ffeats=None, nan_flag=None):
    """EfficientUpdateFormer.forward as one library call (bf16 mode, shipped geometry); workspace: uint8 device tensor.
    With ``coords`` / ``ffeats`` the track and feature update runs inside the flow-head kernel (``delta`` may then be None)."""
    _call("mvt_updateformer_forward", C.addressof(weights), _ptr(_f32c(x)), ldx, n, _ptr(delta), ldd, _ptr(coords), _ptr(ffeats),
          _ptr(nan_flag), _ptr(workspace), workspace.numel(), _stream())


def update_head_bf16(tok, ldt, w0, b0, w2, b2, w4, b4, gn_w, gn_b, wu, bu, coords, ffeats, delta, ldd, rows, hidden, out_dim, nan_flag=None):
    _call("mvt_update_head_bf16", _ptr(tok), ldt, _ptr(w0), _ptr(b0), _ptr(w2), _ptr(b2), _ptr(w4), _ptr(b4), _ptr(gn_w), _ptr(gn_b), _ptr(wu),
          _ptr(bu), _ptr(coords), _ptr(ffeats), _ptr(delta), ldd, rows, hidden, out_dim, _ptr(nan_flag), _stream())


ATTN_TIME, ATTN_FRAME = 1, 2


class BlockAttn(C.Structure):
    """mvt_block_attn."""
    _fields_ = [("kind", C.c_int), ("S", C.c_int), ("n_keys", C.c_int), ("heads", C.c_int), ("dim_head", C.c_int), ("ldq", C.c_int),
                ("ldkv", C.c_int), ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("partials", C.c_void_p), ("n_splits", C.c_int),
                ("defer_pass2", C.c_int), ("ctx", C.c_void_p)]  # (the context form is only driven by the composite updater call)


def attn_block_fused_bf16(x, ldx, kind, S, q, ldq, k, v, ldkv, n_keys, wo, bo, w1, b1, w2, b2, H, nexts, M, Cc, ws=None):
    """``block_fused_bf16`` with the preceding attention inside the kernel (bf16 q / k / v; 6 heads x 48)."""
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16
    at = BlockAttn(kind, S, n_keys, 6, 48, ldq, ldkv, _ptr(q), _ptr(k), _ptr(v), None, 0)
    arr = (BlockNext * max(1, len(nexts)))()
    for i, nx in enumerate(nexts):
        arr[i] = BlockNext(_ptr(nx["w"]), _ptr(nx["b"]), _ptr(nx.get("lnw")), _ptr(nx.get("lnb")), _ptr(nx["y"]), nx["ldw"], nx["N"],
                           nx["ldy"], nx["eps"], *nx.get("rows", (0, 0)), 1 if nx["y"].dtype == torch.bfloat16 else 0)
    _call("mvt_attn_block_fused_bf16", _ptr(x), ldx, C.addressof(at), _ptr(wo), _ptr(bo), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), H,
          C.cast(arr, C.c_void_p), len(nexts), M, Cc, _ptr(ws), _stream())


def track_metrics(gt_tracks, pred_tracks, gt_visible, pred_occluded, query_frame, T, N, D, thresholds, survival_threshold, out):
    """Per-track evaluation metrics; gt_visible / pred_occluded uint8 (T,N), query_frame int32 (N), out (N, 11 + 2K) fp32."""
    assert gt_visible.dtype == torch.uint8 and pred_occluded.dtype == torch.uint8 and query_frame.dtype == torch.int32
    K = len(thresholds)
    th = (C.c_float * K)(*[float(t) for t in thresholds])
    _call("mvt_track_metrics", _ptr(_f32c(gt_tracks)), _ptr(_f32c(pred_tracks)), _ptr(gt_visible), _ptr(pred_occluded), _ptr(query_frame),
          T, N, D, C.cast(th, C.c_void_p), K, float(survival_threshold), _ptr(out), out.shape[1], _stream())


class TokenInputs(C.Structure):
    """mvt_token_inputs."""
    _fields_ = [("coords", C.c_void_p), ("fcorr", C.c_void_p), ("ffeats", C.c_void_p), ("mask_vis", C.c_void_p), ("pos", C.c_void_p),
                ("time_embed", C.c_void_p), ("Fc", C.c_int), ("Cf", C.c_int), ("E", C.c_int)]


def updateformer_forward_tokens(weights: UpdaterWeights, coords, fcorr, Fc, ffeats, Cf, mask_vis, pos, time_embed, E, n, delta, ldd, workspace,
                                upd_coords=None, upd_ffeats=None, nan_flag=None):
    """``updateformer_forward`` with the token rows assembled inside its first kernel (no token matrix)."""
    ti = TokenInputs(_ptr(_f32c(coords)), _ptr(_f32c(fcorr)), _ptr(_f32c(ffeats)), _ptr(_f32c(mask_vis)), _ptr(_f32c(pos)), _ptr(_f32c(time_embed)),
                     Fc, Cf, E)
    _call("mvt_updateformer_forward_tokens", C.addressof(weights), C.addressof(ti), n, _ptr(delta), ldd, _ptr(upd_coords), _ptr(upd_ffeats),
          _ptr(nan_flag), _ptr(workspace), workspace.numel(), _stream())


ENCODER_CONVS = 23


class ConvWeights(C.Structure):
    """mvt_conv_weights."""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class EncoderWeights(C.Structure):
    """mvt_encoder_weights (host struct of device pointers; keep the tensors referenced)."""
    _fields_ = [("latent_dim", C.c_int), ("short_workgroups", C.c_int), ("conv", ConvWeights * ENCODER_CONVS)]


def encoder_workspace_bytes(n, H, W, Cc) -> int:
    return int(_lib.mvt_encoder_workspace_bytes(n, H, W, Cc))


def encoder_forward_rgb(weights: EncoderWeights, rgbs, V, T, img0, n, H, W, out_rows, ldo, workspace):
    """``encoder_forward`` reading images img0 .. img0+n-1 (frame-major) of the planar clip rgbs (V,T,3,H,W) fp32 / uint8 directly."""
    assert rgbs.is_contiguous() and rgbs.dtype in (torch.uint8, torch.float32) and tuple(rgbs.shape) == (V, T, 3, H, W)
    _call("mvt_encoder_forward_rgb", C.addressof(weights), _ptr(rgbs), 1 if rgbs.dtype == torch.uint8 else 0, V, T, img0, n, H, W,
          _ptr(out_rows), ldo, 1 if out_rows.dtype == torch.bfloat16 else 0, _ptr(workspace), workspace.numel(), _stream())


def encoder_forward(weights: EncoderWeights, x4, n, H, W, out_rows, ldo, workspace):
    """BasicEncoder.forward as one library call (bf16 mode): x4 (n,H,W,4) fp32 -> out_rows (n,H/4,W/4,ldo) fp32 or bf16."""
    _call("mvt_encoder_forward", C.addressof(weights), _ptr(_f32c(x4)), n, H, W, _ptr(out_rows), ldo,
          1 if out_rows.dtype == torch.bfloat16 else 0, _ptr(workspace), workspace.numel(), _stream())
