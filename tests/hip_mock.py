"""Torch-CPU stand-ins for the wrappers of mvtracker_amd.hip -- TEST INFRASTRUCTURE ONLY.

The build container has no GPU, so the host-side sequencing of mvtracker_amd.tracker /
.predictor (window bookkeeping, buffer strides, weight packing, launch order) is exercised on
CPU by monkeypatching the ctypes wrappers with the functions below, each of which restates the
documented contract of one entry point of include/mvtracker_hip.h with plain torch ops.  The
product never imports this module; on the GPU box the `-m gpu` tests run the real library.
"""
import math

import torch
import torch.nn.functional as F


def _v(t, rows, cols, ld):
    return torch.as_strided(t, (rows, cols), (ld, 1))


def _act(x, act):
    return [lambda a: a, F.relu, lambda a: F.gelu(a, approximate="tanh"), F.gelu][act](x)


def gemm(A, lda, Wt, ldw, bias, R, ldr, Cm, ldc, M, N, K, act=0):
    assert lda % 4 == 0 and ldw % 32 == 0 and ldw >= K
    a = _v(A, M, K, lda)
    w = _v(Wt, N, ldw, ldw)[:, :K]
    assert float(_v(Wt, N, ldw, ldw)[:, K:].abs().sum()) == 0.0
    y = a @ w.t()
    if bias is not None:
        y = y + bias[:N]
    y = _act(y, act)
    if R is not None:
        y = y + _v(R, M, N, ldr)
    _v(Cm, M, N, ldc).copy_(y)


def _bf(x):
    return x.to(torch.bfloat16).float()


def split_bf16(src, hi, lo, n):
    s = src.reshape(-1)[:n]
    h = s.to(torch.bfloat16)
    hi.reshape(-1)[:n].copy_(h.view(torch.int16))
    if lo is not None:
        lo.reshape(-1)[:n].copy_((s - h.float()).to(torch.bfloat16).view(torch.int16))


def _w_from_bf16(whi, wlo, rows, ld):
    w = torch.as_strided(whi, (rows, ld), (ld, 1)).view(torch.bfloat16).float()
    if wlo is not None:
        w = w + torch.as_strided(wlo, (rows, ld), (ld, 1)).view(torch.bfloat16).float()
    return w


def gemm_bf16(A, lda, Whi, Wlo, ldw, bias, R, ldr, Cm, ldc, M, N, K, act=0):
    assert ldw % 64 == 0 and ldw >= K
    a = _v(A, M, K, lda)
    w = _w_from_bf16(Whi, Wlo, N, ldw)
    assert float(w[:, K:].abs().sum()) == 0.0
    if Wlo is None:
        a = _bf(a)
    y = a @ w[:, :K].t()
    if bias is not None:
        y = y + bias[:N]
    y = _act(y, act)
    if R is not None:
        y = y + _v(R, M, N, ldr)
    _v(Cm, M, N, ldc).copy_(y)


def ln_gemm_bf16(A, lda, ln_w, ln_b, eps, Whi, Wlo, ldw, bias, R, ldr, Cm, ldc, M, N, K, act=0):
    a = F.layer_norm(_v(A, M, K, lda), (K,), ln_w, ln_b, eps).contiguous()
    gemm_bf16(a, K, Whi, Wlo, ldw, bias, R, ldr, Cm, ldc, M, N, K, act)


def pack_frag_bf16(w, ld, N, K, out):
    nb = (N + 31) // 32
    src = torch.zeros(nb * 32, K, dtype=torch.int16)
    src[:N] = torch.as_strided(w, (N, K), (ld, 1))
    out.reshape(-1)[:nb * 32 * K].copy_(src.reshape(nb, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1))


def _unfrag(wf, N, K):
    nb = (N + 31) // 32
    return wf.reshape(-1)[:nb * 32 * K].reshape(nb, K // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(nb * 32, K)[:N].view(torch.bfloat16).float()


def block_fused_bf16(x, ldx, att, ldatt, Ko, wo, ldwo, bo, w1, ldw1, b1, w2, ldw2, b2, H, nexts, M, Cc, ws=None):
    xv = _v(x, M, Cc, ldx)
    if att is not None:
        xv += _bf(_v(att, M, Ko, ldatt).float()) @ _unfrag(wo, Cc, Ko).t() + bo[:Cc]
    hdn = F.gelu(_bf(F.layer_norm(xv, (Cc,), None, None, 1e-6)) @ _unfrag(w1, H, Cc).t() + b1[:H], approximate="tanh")
    xv += _bf(hdn) @ _unfrag(w2, Cc, H).t() + b2[:Cc]
    for nx in nexts:
        Wn = _unfrag(nx["w"], nx["N"], Cc)
        a = _bf(F.layer_norm(xv, (Cc,), nx.get("lnw"), nx.get("lnb"), nx["eps"]))
        lo, hi = nx.get("rows", (0, 0))
        hi = hi or M
        _v(nx["y"], hi, nx["N"], nx["ldy"])[lo:hi] = (a @ Wn.t() + nx["b"][:nx["N"]])[lo:hi].to(nx["y"].dtype)


def ln_proj_bf16(x, ldx, nexts, M, Cc):
    xv = _v(x, M, Cc, ldx)
    for nx in nexts:
        Wn = _unfrag(nx["w"], nx["N"], Cc)
        a = _bf(F.layer_norm(xv, (Cc,), nx.get("lnw"), nx.get("lnb"), nx["eps"]))
        lo, hi = nx.get("rows", (0, 0))
        hi = hi or M
        _v(nx["y"], hi, nx["N"], nx["ldy"])[lo:hi] = (a @ Wn.t() + nx["b"][:nx["N"]])[lo:hi].to(nx["y"].dtype)


def mlp_fused_bf16(x, ldx, w1, ldw1, b1, w2, ldw2, b2, M, Cc, H, eps):
    xv = _v(x, M, Cc, ldx)
    W1 = torch.as_strided(w1, (H, ldw1), (ldw1, 1)).view(torch.bfloat16).float()[:, :Cc]
    W2 = torch.as_strided(w2, (Cc, ldw2), (ldw2, 1)).view(torch.bfloat16).float()[:, :H]
    hdn = F.gelu(_bf(F.layer_norm(xv, (Cc,), None, None, eps)) @ W1.t() + b1[:H], approximate="tanh")
    xv += _bf(hdn) @ W2.t() + b2[:Cc]


def conv2d_stat_slots(H, W, Cin, KH, KW, stride, pad, split=False):
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if not split and Cin % 32 == 0 and KH == KW and ((KH == 3 and pad == 1) or (KH == 1 and pad == 0)):
        return Ho * ((Wo + 31) // 32)
    if KH == 3 and KW == 3 and stride == 1 and pad == 1 and Cin % 32 == 0:
        return ((Ho + 7) // 8) * ((Wo + 15) // 16) * 4
    if not split and Cin == 4 and KH == 7 and KW == 7 and stride == 2 and pad == 3:
        return Ho * ((Wo + 31) // 32)
    return Ho * Wo // 32 if (Ho * Wo) % 256 == 0 else 0


def conv3x3s2_down_bf16(x, w3, b3, wd, bd, out3, outd, n, H, W, Cin, Cout, ldo, part3=None, partd=None):
    """conv1 + downsample[0] of a strided ResidualBlock: the two convolutions of the same input (mock: two calls)."""
    conv2d_bf16(x, w3, None, b3, out3, n, H, W, Cin, Cout, 3, 3, 2, 1, ldo, out_partial=part3)
    conv2d_bf16(x, wd, None, bd, outd, n, H, W, Cin, Cout, 1, 1, 2, 0, ldo, out_partial=None)
    if partd is not None:  # (the fused kernel cuts BOTH outputs into the 3x3 kernel's slots; the mock: whole-image sum in slot 0)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        slots = conv2d_stat_slots(H, W, Cin, 3, 3, 2, 1)
        y = torch.as_strided(outd, (n, Ho * Wo, Cout), (Ho * Wo * ldo, ldo, 1)).float()
        pp = partd.reshape(-1)[:n * slots * Cout * 2].reshape(n, slots, Cout, 2)
        pp.zero_()
        pp[:, 0, :, 0] = y.sum(1)
        pp[:, 0, :, 1] = (y * y).sum(1)


def conv2d_bf16(x, wt_hi, wt_lo, bias, out, n, H, W, Cin, Cout, KH, KW, stride, pad, ldo, act=0, in_stats=None, out_partial=None, short_wg=False):
    K = KH * 32 if Cin == 4 else KH * KW * Cin
    ld = (K + 63) // 64 * 64
    w = _w_from_bf16(wt_hi, wt_lo, Cout, ld).contiguous()
    xin = x
    if in_stats is not None:
        st = in_stats.reshape(n, 1, Cin, 2)
        xin = F.relu((torch.as_strided(x, (n, H * W, Cin), (H * W * Cin, Cin, 1)).float() - st[..., 0]) * st[..., 1]).contiguous()
    xin = xin if wt_lo is not None else _bf(xin.float())
    conv2d(xin, w, bias, out, n, H, W, Cin, Cout, KH, KW, stride, pad, ldo, act)
    if out_partial is not None:  # the mock puts the whole image sum into slot 0
        slots = conv2d_stat_slots(H, W, Cin, KH, KW, stride, pad, wt_lo is not None)
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        y = torch.as_strided(out, (n, Ho * Wo, Cout), (Ho * Wo * ldo, ldo, 1)).float()
        pp = out_partial.reshape(-1)[:n * slots * Cout * 2].reshape(n, slots, Cout, 2)
        pp.zero_()
        pp[:, 0, :, 0] = y.sum(1)
        pp[:, 0, :, 1] = (y * y).sum(1)


def instnorm_finish_slots(partial, slots, mean_rstd, n, HW, Cc):
    pp = partial.reshape(-1)[:n * slots * Cc * 2].reshape(n, slots, Cc, 2).double().sum(1)
    mean = pp[..., 0] / HW
    var = (pp[..., 1] / HW - mean * mean).clamp_min(0)
    st = mean_rstd.reshape(n, Cc, 2)
    st[..., 0] = mean.float()
    st[..., 1] = (1.0 / torch.sqrt(var + 1e-5)).float()


def conv2d(x, wt, bias, out, n, H, W, Cin, Cout, KH, KW, stride, pad, ldo, act=0):
    xi = torch.as_strided(x, (n, H, W, Cin), (H * W * Cin, W * Cin, Cin, 1)).permute(0, 3, 1, 2).float()
    ld = ((KH * 32 if Cin == 4 else KH * KW * Cin) + 63) // 64 * 64
    if Cin == 4:
        w = torch.as_strided(wt, (Cout, KH, 8, 4), (ld, 32, 4, 1))[:, :, :KW, :].permute(0, 3, 1, 2)
    else:
        w = torch.as_strided(wt, (Cout, KH, KW, Cin), (ld, KW * Cin, Cin, 1)).permute(0, 3, 1, 2)
    y = _act(F.conv2d(xi, w, bias, stride=stride, padding=pad), act)
    Ho, Wo = y.shape[-2:]
    torch.as_strided(out, (n, Ho, Wo, Cout), (Ho * Wo * ldo, Wo * ldo, ldo, 1)).copy_(y.permute(0, 2, 3, 1))


def rgb_to_nhwc4(rgbs, out, V, T, H, W, t0, nt):
    x = 2 * (rgbs[:, t0:t0 + nt].float() / 255.0) - 1.0  # (V,nt,3,H,W)
    o = torch.as_strided(out, (nt, V, H, W, 4), (V * H * W * 4, H * W * 4, W * 4, 4, 1))
    o[..., :3] = x.permute(1, 0, 3, 4, 2)
    o[..., 3] = 0


def rgb_images_to_nhwc4(rgbs, out, V, T, H, W, img0, nimg):
    o = torch.as_strided(out, (nimg, H, W, 4), (H * W * 4, W * 4, 4, 1))
    for i in range(nimg):
        t, v = divmod(img0 + i, V)
        o[i, ..., :3] = (2 * (rgbs[v, t].float() / 255.0) - 1.0).permute(1, 2, 0)
        o[i, ..., 3] = 0


def resize_nearest(x, out, planes, Hi, Wi, Ho, Wo):
    y = F.interpolate(x.reshape(1, planes, Hi, Wi), (Ho, Wo), mode="nearest")
    out.reshape(-1)[:planes * Ho * Wo].copy_(y.reshape(-1))


def instnorm_stats(x, ldx, partial, mean_rstd, n, HW, Cc):
    xv = torch.as_strided(x, (n, HW, Cc), (HW * ldx, ldx, 1)).double()
    mean = xv.mean(1)
    var = (xv * xv).mean(1) - mean * mean
    st = torch.as_strided(mean_rstd, (n, Cc, 2), (Cc * 2, 2, 1))
    st[..., 0] = mean.float()
    st[..., 1] = (1.0 / torch.sqrt(var.clamp_min(0) + 1e-5)).float()


def instnorm_apply(x, mean_rstd, skip, skip_stats, y, n, HW, Cc, skip_relu=False):
    xv = torch.as_strided(x, (n, HW, Cc), (HW * Cc, Cc, 1)).float()
    st = torch.as_strided(mean_rstd, (n, 1, Cc, 2), (Cc * 2, 0, 2, 1))
    o = F.relu((xv - st[..., 0]) * st[..., 1])
    if skip is not None:
        k = torch.as_strided(skip, (n, HW, Cc), (HW * Cc, Cc, 1)).float()
        if skip_stats is not None:
            ks = torch.as_strided(skip_stats, (n, 1, Cc, 2), (Cc * 2, 0, 2, 1))
            k = (k - ks[..., 0]) * ks[..., 1]
            if skip_relu:
                k = F.relu(k)
        o = F.relu(k + o)
    torch.as_strided(y, (n, HW, Cc), (HW * Cc, Cc, 1)).copy_(o)


def resize_bilinear_ac(src, dst, n, Hs, Ws, Cc, Hd, Wd, ldd, c_off):
    s = torch.as_strided(src, (n, Hs, Ws, Cc), (Hs * Ws * Cc, Ws * Cc, Cc, 1)).permute(0, 3, 1, 2).float()
    y = F.interpolate(s, (Hd, Wd), mode="bilinear", align_corners=True)
    torch.as_strided(dst, (n, Hd, Wd, Cc), (Hd * Wd * ldd, Wd * ldd, ldd, 1), dst.storage_offset() + c_off).copy_(y.permute(0, 2, 3, 1))


def concat_resize_bilinear_ac(srcs, dims, dst, n, Hd, Wd, ldd):
    off = 0
    for t, (hs_, ws_, c_) in zip(srcs, dims):
        resize_bilinear_ac(t, dst, n, hs_, ws_, c_, Hd, Wd, ldd, off)
        off += c_


def invert_cameras(intrs, extrs, kinv, einv, n):
    k = torch.inverse(intrs.reshape(n, 3, 3).double())
    e = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1)
    e[:, :3] = extrs.reshape(n, 3, 4).double()
    kinv.reshape(n, 9).copy_(k.reshape(n, 9).float())
    einv.reshape(n, 12).copy_(torch.inverse(e)[:, :3].reshape(n, 12).float())


def depth_subsample(depths, out, V, T, H, W, s):
    d = depths.reshape(V, T, H, W)[:, :, ::s, ::s][:, :, :H // s, :W // s]
    out.reshape(T, V, H // s, W // s).copy_(d.permute(1, 0, 2, 3))


def avgpool2(x, out, n, h, w, Cc):
    xi = torch.as_strided(x, (n, h, w, Cc), (h * w * Cc, w * Cc, Cc, 1)).permute(0, 3, 1, 2)
    y = F.avg_pool2d(xi, 2, stride=2).permute(0, 2, 3, 1)
    torch.as_strided(out, tuple(y.shape), (y.shape[1] * y.shape[2] * Cc, y.shape[2] * Cc, Cc, 1)).copy_(y)


def unproject(depth_s, kinv, einv, xyz, V, T, hs, ws, stride, level):
    f = 1 << level
    h, w = hs >> level, ws >> level
    d = depth_s.reshape(T, V, hs, ws)[:, :, ::f, ::f][:, :, :h, :w]
    st = stride * f
    ys = (torch.arange(h) + 0.5) * st - 0.5
    xs = (torch.arange(w) + 0.5) * st - 0.5
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    pix = torch.stack([gx, gy, torch.ones_like(gx)], -1)
    K = kinv.reshape(V, T, 3, 3).permute(1, 0, 2, 3)
    E = einv.reshape(V, T, 3, 4).permute(1, 0, 2, 3)
    cam = torch.einsum("tvij,hwj->tvhwi", K, pix) * d[..., None]
    world = torch.einsum("tvij,tvhwj->tvhwi", E[..., :3], cam) + E[:, :, None, None, :, 3]
    o = xyz.reshape(T, V, h, w, 4)
    o[..., :3] = world
    o[..., 3] = 0


def _d2(ref, q):
    dx = ref[None, :, 0] - q[:, None, 0]
    dy = (ref[None, :, 1] - q[:, None, 1]).double()
    dz = (ref[None, :, 2] - q[:, None, 2]).double()
    acc = (dy * dy + (dx * dx).double()).float()
    return (dz * dz + acc.double()).float()


def tile_aabb(xyz, Pn, T, box, grid=(0, 0)):
    box.zero_()  # culling only prunes: the mock scan ignores the boxes


def knn_scan(xyz, Pn, coords, N, S, frame0, frame_step, T, K, nseg, keys, seed_idx=None, seed_k=0, seed_dims=(0, 0, 0, 0), box=None,
             grid=(0, 0)):
    # the seed / boxes only prune; the result is the exact kNN either way.  Check the seed contract: valid distinct indices.
    if seed_idx is not None:
        si = seed_idx.reshape(N, S, seed_k).long()
        cw, ch, fw, fh = seed_dims
        if cw > 0:
            v, rem = si // (cw * ch), si % (cw * ch)
            si = (v * fh + 2 * (rem // cw)) * fw + 2 * (rem % cw)
        assert seed_k >= K and int(si.min()) >= 0 and int(si.max()) < Pn
        assert all(len(set(row.tolist())) == seed_k for row in si.reshape(-1, seed_k)[:64])
    X = torch.as_strided(xyz, (T, Pn, 4), (Pn * 4, 4, 1))
    c = torch.as_strided(coords, (N, S, 3), (S * 3, 3, 1))
    kv = torch.as_strided(keys, (N, S, nseg, K), (S * nseg * K, nseg * K, K, 1))
    per = ((Pn + 63) // 64 + nseg - 1) // nseg * 64
    kv.fill_(torch.iinfo(torch.int64).max)  # pads a short segment (sorts last, like the library's KEY_MAX)
    for s in range(S):
        f = min(frame0 + s * frame_step, T - 1)
        d2 = _d2(X[f, :, :3], c[:, s])
        key = (d2.view(torch.int32).to(torch.int64) << 32) | torch.arange(Pn)[None]
        for g in range(nseg):
            seg = key[:, g * per:min(Pn, (g + 1) * per)]
            kk = min(K, seg.shape[1])
            kv[:, s, g, :kk] = torch.topk(seg, kk, dim=1, largest=False, sorted=True).values


def knn_scan_levels(levels, coords, N, S, frame0, frame_step, T, K, seed_k=0):
    for lv in levels:
        knn_scan(lv["xyz"], lv["P"], coords, N, S, frame0, frame_step, T, K, lv["nseg"], lv["keys"], seed_idx=lv.get("seed_idx"),
                 seed_k=seed_k, box=lv.get("box"), grid=lv.get("grid", (0, 0)))


def tile_group_aabb(box, Pn, T, group_box):
    nt = (Pn + 63) // 64
    for g in range((nt + 63) // 64):
        b = box[:, 64 * g:64 * g + 64]
        group_box[:, g, 0:3] = b[..., 0:3].amin(1)
        group_box[:, g, 3] = b[..., 3].sum(1)
        group_box[:, g, 4:7] = b[..., 4:7].amax(1)
        group_box[:, g, 7] = 0


def knn_search(xyz, Pn, coords, N, S, frame0, frame_step, T, K, idx_out, box, grid=(0, 0), gbox=None, seed_idx=None, seed_k=0,
               seed_dims=(0, 0, 0, 0)):
    keys = torch.empty(N * S * K, dtype=torch.int64, device=coords.device)
    knn_scan(xyz, Pn, coords, N, S, frame0, frame_step, T, K, 1, keys, seed_idx=seed_idx, seed_k=seed_k, seed_dims=seed_dims, box=box, grid=grid)
    knn_merge(keys, N, S, K, 1, Pn, idx_out)


def knn_search_levels(levels, coords, N, S, frame0, frame_step, T, K, seed_k):
    for lv in levels:
        assert (lv.get("seed_idx") is not None) == (seed_k > 0)
        keys = torch.empty(N * S * K, dtype=torch.int64, device=coords.device)
        knn_scan(lv["xyz"], lv["P"], coords, N, S, frame0, frame_step, T, K, 1, keys, seed_idx=lv.get("seed_idx"), seed_k=seed_k,
                 box=lv.get("box"), grid=lv.get("grid", (0, 0)))
        knn_merge(keys, N, S, K, 1, lv["P"], lv["idx_out"])


def knn_merge_levels(levels, N, S, K):
    for lv in levels:
        knn_merge(lv["keys"], N, S, K, lv["nseg"], lv["P"], lv["idx_out"])


def knn_merge(keys, N, S, K, nseg, Pn, idx_out):
    kv = torch.as_strided(keys, (N, S, nseg * K), (S * nseg * K, nseg * K, 1))
    idx = (torch.sort(kv, dim=2).values[:, :, :K] & 0xFFFFFFFF).clamp(max=Pn - 1)
    idx_out.reshape(N, S, K).copy_(idx.int())


def corr_gather_dot(xyz_l, fvec_l, P_l, idx_l, Cc, targets, coords, N, S, frame0, frame_step, T, K, out, ldo, o_off):
    tg = torch.as_strided(targets, (N, S, Cc), (S * Cc, Cc, 1))
    c = torch.as_strided(coords, (N, S, 3), (S * 3, 3, 1))
    for lvl, (xyz, fvec, Pn, idx_t) in enumerate(zip(xyz_l, fvec_l, P_l, idx_l)):
        X = torch.as_strided(xyz, (T, Pn, 4), (Pn * 4, 4, 1))
        Fv = torch.as_strided(fvec, (T, Pn, Cc), (Pn * Cc, Cc, 1))
        idx = idx_t.reshape(N, S, K).long()
        o = torch.as_strided(out, (N, S, K, 4), (S * ldo, ldo, 4, 1), out.storage_offset() + o_off + lvl * 4 * K)
        for s in range(S):
            f = min(frame0 + s * frame_step, T - 1)
            nf = Fv[f][idx[:, s]]
            o[:, s, :, 0] = torch.einsum("nc,nkc->nk", tg[:, s], nf) / math.sqrt(Cc)
            o[:, s, :, 1:] = X[f][idx[:, s]][..., :3] - c[:, s, None]


def corr_gather_dot_opts(xyz_l, fvec_l, P_l, idx_l, Cc, targets, coords, N, S, frame0, frame_step, T, K, groups, add_offset, add_xyz, out, ldo,
                         o_off):
    OW = groups + 3 * int(add_offset) + 3 * int(add_xyz)
    tg = torch.as_strided(targets, (N, S, Cc), (S * Cc, Cc, 1))
    c = torch.as_strided(coords, (N, S, 3), (S * 3, 3, 1))
    for lvl, (xyz, fvec, Pn, idx_t) in enumerate(zip(xyz_l, fvec_l, P_l, idx_l)):
        X = torch.as_strided(xyz, (T, Pn, 4), (Pn * 4, 4, 1))
        Fv = torch.as_strided(fvec, (T, Pn, Cc), (Pn * Cc, Cc, 1))
        idx = idx_t.reshape(N, S, K).long()
        o = torch.as_strided(out, (N, S, K, OW), (S * ldo, ldo, OW, 1), out.storage_offset() + o_off + lvl * OW * K)
        for s in range(S):
            f = min(frame0 + s * frame_step, T - 1)
            nf = Fv[f][idx[:, s]].float()
            o[:, s, :, :groups] = torch.einsum("ngc,nkgc->nkg", tg[:, s].reshape(N, groups, -1), nf.reshape(N, K, groups, -1)) / math.sqrt(Cc / groups)
            nx = X[f][idx[:, s]][..., :3]
            j = groups
            if add_offset:
                o[:, s, :, j:j + 3] = nx - c[:, s, None]
                j += 3
            if add_xyz:
                o[:, s, :, j:j + 3] = nx


def knn1_gather(fvec, Pn, Cc, keys, n, nseg, frame, feat_out, idx_out=None):
    idx = keys.reshape(n, nseg).min(1).values & 0xFFFFFFFF
    Fv = torch.as_strided(fvec, (frame + 1, Pn, Cc), (Pn * Cc, Cc, 1))
    feat_out.reshape(n, Cc).copy_(Fv[frame][idx])
    if idx_out is not None:
        idx_out.reshape(n).copy_(idx.int())


def pos_embed(coords, N, S, D, dim_padded, pos, omega=None):
    import numpy as np
    A = dim_padded // 3
    omega = 1.0 / 10000 ** (np.arange(A // 2, dtype=np.float64) / (A / 2.0))
    c0 = torch.as_strided(coords, (N, 3), (S * 3, 1)).double().numpy()
    e = []
    for a in range(3):
        o = c0[:, a:a + 1] * omega[None]
        e += [np.sin(o), np.cos(o)]
    pos.reshape(N, D).copy_(torch.from_numpy(np.concatenate(e, 1)[:, :D]).float())


def token_assemble(coords, fcorr, Fc, ffeats, Cc, mask_vis, pos, time_embed, N, S, E, x, ldx):
    c = coords.reshape(N, S, 3)
    fl = c - c[:, :1]
    div = (torch.arange(0, E, 2, dtype=torch.float32) * (1000.0 / E)).reshape(1, 1, E // 2)
    parts = []
    for a in range(3):
        pe = torch.zeros(N, S, E)
        pe[:, :, 0::2] = torch.sin(fl[:, :, a:a + 1] * div)
        pe[:, :, 1::2] = torch.cos(fl[:, :, a:a + 1] * div)
        parts.append(pe)
    D = 3 * E + 3 + Fc + Cc + 2
    t = torch.cat(parts + [fl, fcorr.reshape(N, S, Fc), ffeats.reshape(N, S, Cc), mask_vis.reshape(N, S, 2)], 2)
    t = t + pos.reshape(N, 1, D) + time_embed.reshape(1, S, D)
    torch.as_strided(x, (N * S, D), (ldx, 1)).copy_(t.reshape(N * S, D))
    if ldx > D:
        torch.as_strided(x, (N * S, ldx - D), (ldx, 1), x.storage_offset() + D).zero_()


def delta_split(delta, ldd, gw, gb, coords, dn, rows, Cc, nan_flag=None):
    d = _v(delta, rows, 3 + Cc, ldd)
    c = coords.reshape(-1)[:rows * 3].reshape(rows, 3)
    c += d[:, :3]
    dn.reshape(-1)[:rows * Cc].reshape(rows, Cc).copy_(F.group_norm(d[:, 3:], 1, gw, gb, 1e-5))
    if nan_flag is not None and bool(torch.isnan(c).any()):
        nan_flag.fill_(1)


def rowdot(x, ldx, w, b, out, rows, Cc):
    out.reshape(-1)[:rows].copy_(_v(x, rows, Cc, ldx) @ w[:Cc] + b[0])


def layernorm(x, ldx, w, b, y, ldy, rows, Cc, eps):
    _v(y, rows, Cc, ldy).copy_(F.layer_norm(_v(x, rows, Cc, ldx), (Cc,), w, b, eps))


def attention(q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, o, ldo, groups, nq, nk, heads, dh):
    Q = torch.as_strided(q, (groups, nq, heads, dh), (q_gs * ldq, q_is * ldq, dh, 1)).permute(0, 2, 1, 3).float()
    Kt = torch.as_strided(k, (groups, nk, heads, dh), (k_gs * ldkv, k_is * ldkv, dh, 1)).permute(0, 2, 1, 3).float()
    Vt = torch.as_strided(v, (groups, nk, heads, dh), (k_gs * ldkv, k_is * ldkv, dh, 1)).permute(0, 2, 1, 3).float()
    y = F.scaled_dot_product_attention(Q, Kt, Vt).permute(0, 2, 1, 3)
    torch.as_strided(o, (groups, nq, heads, dh), (q_gs * ldo, q_is * ldo, dh, 1)).copy_(y)


def attention_ws_floats(groups, nq, heads):
    return 4 * groups * heads * ((nq + 63) // 64) * 64 * 68


def attention_bf16(q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, o, ldo, groups, nq, nk, heads, dh, ws=None):
    attention(q, ldq, q_gs, q_is, k, v, ldkv, k_gs, k_is, o, ldo, groups, nq, nk, heads, dh)


def broadcast_rows(v, x, ld, n, S, Cc):
    torch.as_strided(x, (n, S, Cc), (S * ld, ld, 1)).copy_(v.reshape(n, 1, Cc).expand(n, S, Cc))


def window_corr(fmap, targets, coords, out, BS, N, Cc, h, w, level, radius, ldo, o_off):
    raise NotImplementedError("window_corr is only exercised on the GPU")


def window_prepare(qxyz, qt, feat_init, prev_coords, prev_vis, n, p0, S, Cc, w, T, coords, mask_vis, ffeats):
    half = S // 2
    ffeats.reshape(n, S, Cc).copy_(feat_init.reshape(-1, Cc)[:n, None, :].expand(n, S, Cc))
    c = coords.reshape(n, S, 3)
    c.copy_(qxyz.reshape(-1, 3)[:n, None, :].expand(n, S, 3))
    vis = torch.full((n, S), 10.0)
    if p0 > 0:
        sp = [half + s_ if s_ < half else S - 1 for s_ in range(S)]
        c[:p0] = prev_coords.reshape(-1, S, 3)[:p0][:, sp]
        vis[:p0] = prev_vis.reshape(-1, S)[:p0][:, sp]
    s_local = min(S, T - w)
    f = torch.tensor([w + min(s_, s_local - 1) for s_ in range(S)])
    on = f[None, :] >= qt.reshape(-1)[:n, None]
    on[:p0] &= ~(f[None, :] < w + half)
    mv = mask_vis.reshape(n, S, 2)
    mv[..., 0] = on.float()
    mv[..., 1] = vis


def window_store(coords, vis, order, n, S, w, T, N, traj, vis_logit, vis_prob):
    s_local = min(S, T - w)
    o = order.reshape(-1)[:n]
    traj.reshape(T, N, 3)[w:w + s_local, o] = coords.reshape(n, S, 3)[:, :s_local].permute(1, 0, 2)
    vis_logit.reshape(T, N)[w:w + s_local, o] = vis.reshape(n, S)[:, :s_local].t()
    vis_prob.reshape(T, N)[w:w + s_local, o] = torch.sigmoid(vis.reshape(n, S)[:, :s_local].t())


def require_device(t):
    return None


def install(monkeypatch):
    """Patch mvtracker_amd.hip so that the host code runs on CPU tensors."""
    import sys
    from mvtracker_amd import hip
    me = sys.modules[__name__]
    for name in ("gemm conv2d split_bf16 gemm_bf16 conv2d_stat_slots conv2d_bf16 instnorm_finish_slots ln_gemm_bf16 pack_frag_bf16 block_fused_bf16 ln_proj_bf16 mlp_fused_bf16 rgb_to_nhwc4 rgb_images_to_nhwc4 resize_nearest instnorm_stats instnorm_apply resize_bilinear_ac concat_resize_bilinear_ac invert_cameras "
                 "attention_ws_floats depth_subsample avgpool2 unproject tile_aabb knn_scan knn_merge knn_scan_levels knn_search_levels knn_search tile_group_aabb knn_merge_levels corr_gather_dot corr_gather_dot_opts knn1_gather pos_embed token_assemble delta_split "
                 "rowdot layernorm attention attention_bf16 broadcast_rows window_corr window_prepare window_store require_device").split():
        monkeypatch.setattr(hip, name, getattr(me, name))
    monkeypatch.setattr(hip, "COMPOSITE", False)  # the per-kernel sequencing is what these tests exercise
