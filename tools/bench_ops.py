"""Micro-benchmarks of individual library kernels at the updater / encoder shapes of the C3 benchmark (GPU box).
    python tools/bench_ops.py [mlp] [gemm] [attn] [knn] [conv]"""
import os
import sys
import math
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda:0"
which = set(sys.argv[1:]) or {"mlp", "gemm", "attn", "knn", "conv"}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def split(w, lo):
    hi = torch.empty(w.shape, device=dev, dtype=torch.int16)
    l = torch.empty(w.shape, device=dev, dtype=torch.int16) if lo else None
    hip.split_bf16(w, hi, l, w.numel())
    return hi, l


if "mlp" in which:
    C, H = 256, 1024
    W1 = torch.randn(H, C, device=dev) / 16
    W2 = torch.randn(C, H, device=dev) / 32
    b1, b2 = torch.randn(H, device=dev), torch.randn(C, device=dev)
    h1, _ = split(W1, False)
    h2, _ = split(W2, False)
    for M in (13056, 12288, 768):
        x = torch.randn(M, C, device=dev)
        xn = torch.empty_like(x)
        hb = torch.empty(M, H, device=dev)
        t_f = timeit(lambda: hip.mlp_fused_bf16(x, C, h1, C, b1, h2, H, b2, M, C, H, 1e-6))

        def unfused():
            hip.layernorm(x, C, None, None, xn, C, M, C, 1e-6)
            hip.gemm_bf16(xn, C, h1, None, C, b1, None, 0, hb, H, M, H, C, 2)
            hip.gemm_bf16(hb, H, h2, None, H, b2, x, C, x, C, M, C, H, 0)

        t_u = timeit(unfused)
        fl = 4 * M * C * H
        print(f"mlp M={M}: fused {t_f:.1f} us ({fl / t_f / 1e6:.0f} TF/s)  unfused LN+fc1+fc2 {t_u:.1f} us ({fl / t_u / 1e6:.0f} TF/s)")

if "gemm" in which:
    for (M, N, K, act) in ((13056, 864, 256, 0), (13056, 256, 288, 0), (12288, 256, 581, 0), (12288, 288, 256, 0), (12288, 576, 256, 0),
                           (768, 864, 256, 0), (768, 256, 1024, 0), (13056, 1024, 256, 2), (13056, 256, 1024, 0), (12288, 131, 256, 1)):
        lda = (K + 3) // 4 * 4
        A = torch.randn(M, lda, device=dev)
        Kp = (K + 63) // 64 * 64
        W = torch.zeros(N, Kp, device=dev)
        W[:, :K] = torch.randn(N, K, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        Cm = torch.empty(M, N, device=dev)
        hi, lo = split(W, True)
        r = {}
        r["fp32"] = timeit(lambda: hip.gemm(A, lda, W, Kp, b, None, 0, Cm, N, M, N, K, act))
        r["bf16x3"] = timeit(lambda: hip.gemm_bf16(A, lda, hi, lo, Kp, b, None, 0, Cm, N, M, N, K, act))
        r["bf16"] = timeit(lambda: hip.gemm_bf16(A, lda, hi, None, Kp, b, None, 0, Cm, N, M, N, K, act))
        fl = 2 * M * N * K
        print(f"gemm M={M} N={N} K={K} act={act}: " + "  ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TF/s)" for k, v in r.items()))

if "attn" in which:
    n, nv, S, Hh, dh = 1024, 64, 12, 6, 48
    inner = Hh * dh
    M = (n + nv) * S
    Mp = n * S
    qkv = torch.randn(M, 3 * inner, device=dev)
    out = torch.empty(M, inner, device=dev)
    print("attn time  %.1f us" % timeit(lambda: hip.attention(qkv, 3 * inner, S, 1, qkv[:, inner:], qkv[:, 2 * inner:], 3 * inner, S, 1, out, inner, n + nv, S, S, Hh, dh)))
    print("attn v2p   %.1f us" % timeit(lambda: hip.attention(qkv[Mp:], 3 * inner, 1, S, qkv[:Mp, inner:], qkv[:Mp, 2 * inner:], 3 * inner, 1, S, out[Mp:], inner, S, nv, n, Hh, dh)))
    print("attn vself %.1f us" % timeit(lambda: hip.attention(qkv[Mp:], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, out[Mp:], inner, S, nv, nv, Hh, dh)))
    aws = torch.empty(hip.attention_ws_floats(S, nv, Hh), device=dev)
    print("attn_bf16 v2p key-split %.1f us" % timeit(lambda: hip.attention_bf16(qkv[Mp:], 3 * inner, 1, S, qkv[:Mp, inner:], qkv[:Mp, 2 * inner:], 3 * inner, 1, S, out[Mp:], inner, S, nv, n, Hh, dh, ws=aws)))
    print("attn_bf16 time %.1f us" % timeit(lambda: hip.attention_bf16(qkv, 3 * inner, S, 1, qkv[:, inner:], qkv[:, 2 * inner:], 3 * inner, S, 1, out, inner, n + nv, S, S, Hh, dh)))
    for nm, f in (("v2p", lambda: hip.attention_bf16(qkv[Mp:], 3 * inner, 1, S, qkv[:Mp, inner:], qkv[:Mp, 2 * inner:], 3 * inner, 1, S, out[Mp:], inner, S, nv, n, Hh, dh)),
                  ("vself", lambda: hip.attention_bf16(qkv[Mp:], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, out[Mp:], inner, S, nv, nv, Hh, dh)),
                  ("p2v", lambda: hip.attention_bf16(qkv[:Mp], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, out[:Mp], inner, S, n, nv, Hh, dh))):
        print("attn_bf16 %s %.1f us" % (nm, timeit(f)))
    print("attn p2v   %.1f us" % timeit(lambda: hip.attention(qkv[:Mp], 3 * inner, 1, S, qkv[Mp:, inner:], qkv[Mp:, 2 * inner:], 3 * inner, 1, S, out[:Mp], inner, S, n, nv, Hh, dh)))

if "knn" in which:
    N, S, K, T = 1024, 12, 16, 12
    for (P, nseg, hw) in ((65536, 4, 128), (16384, 2, 64), (4096, 1, 32), (1024, 1, 16)):
        V = 4
        ys, xs = torch.meshgrid(torch.arange(hw).float(), torch.arange(hw).float(), indexing="ij")
        base = torch.stack([xs * 0.03, ys * 0.03, torch.zeros_like(xs)], -1).reshape(1, 1, hw * hw, 3)
        xyz = torch.zeros(T, P, 4, device=dev)
        xyz[..., :3] = (base + torch.rand(T, V, hw * hw, 3) * 0.01 + torch.arange(V).view(1, V, 1, 1) * 0.007).reshape(T, P, 3).to(dev)
        q = (torch.rand(N, S, 3) * torch.tensor([hw * 0.03, hw * 0.03, 0.01])).to(dev)
        keys = torch.empty(N * S * nseg * K, device=dev, dtype=torch.int64)
        idx = torch.empty(N, S, K, device=dev, dtype=torch.int32)
        t0 = timeit(lambda: hip.knn_scan(xyz, P, q, N, S, 0, 1, T, K, nseg, keys), iters=5, warm=1)
        hip.knn_merge(keys, N, S, K, nseg, P, idx)
        q2 = q + torch.randn_like(q) * 0.002
        t1 = timeit(lambda: hip.knn_scan(xyz, P, q2, N, S, 0, 1, T, K, nseg, keys, seed_idx=idx, seed_k=K), iters=5, warm=1)
        box = torch.empty(T, (P + 63) // 64, 8, device="cuda")
        hip.tile_aabb(xyz, P, T, box)
        t2 = timeit(lambda: hip.knn_scan(xyz, P, q2, N, S, 0, 1, T, K, nseg, keys, seed_idx=idx, seed_k=K, box=box), iters=5, warm=1)
        boxp = torch.empty(T, (P + 63) // 64, 8, device="cuda")
        hip.tile_aabb(xyz, P, T, boxp, (hw, hw))
        t3 = timeit(lambda: hip.knn_scan(xyz, P, q2, N, S, 0, 1, T, K, nseg, keys, seed_idx=idx, seed_k=K, box=boxp, grid=(hw, hw)), iters=5, warm=1)
        t4 = timeit(lambda: hip.knn_scan(xyz, P, q2, N, S, 0, 1, T, K, 1, keys, seed_idx=idx, seed_k=K, box=boxp, grid=(hw, hw)), iters=5, warm=1)
        t5 = timeit(lambda: hip.knn_scan(xyz, P, q, N, S, 0, 1, T, K, nseg, keys, box=boxp, grid=(hw, hw)), iters=5, warm=1)
        print(f"  seeded + culling: linear tiles {t2:.1f} us, 8x8 patches {t3:.1f} us (nseg=1: {t4:.1f} us); unseeded + patches {t5:.1f} us")
        pairs = N * S * P
        print(f"knn P={P} nseg={nseg}: unseeded {t0:.0f} us ({pairs / t0 / 1e3:.0f} Gpair/s)  seeded {t1:.0f} us ({pairs / t1 / 1e3:.0f} Gpair/s)")

if "conv" in which:
    for (n, Hh, Ww, Cin, Cout, k, s, p) in ((16, 256, 256, 64, 64, 3, 1, 1), (16, 128, 128, 96, 96, 3, 1, 1), (16, 128, 128, 416, 256, 3, 1, 1),
                                            (16, 256, 256, 64, 96, 3, 2, 1), (16, 64, 64, 128, 128, 3, 1, 1), (16, 512, 512, 4, 64, 7, 2, 3)):
        x = torch.randn(n, Hh, Ww, Cin, device=dev)
        K = k * 32 if Cin == 4 else k * k * Cin
        Kp = (K + 63) // 64 * 64
        W = torch.zeros(Cout, Kp, device=dev)
        W[:, :K] = torch.randn(Cout, K, device=dev) / math.sqrt(K)
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (Hh + 2 * p - k) // s + 1, (Ww + 2 * p - k) // s + 1
        out = torch.empty(n, Ho, Wo, Cout, device=dev)
        hi, lo = split(W, True)
        r = {}
        r["bf16x3"] = timeit(lambda: hip.conv2d_bf16(x, hi, lo, b, out, n, Hh, Ww, Cin, Cout, k, k, s, p, Cout), iters=5, warm=1)
        r["bf16"] = timeit(lambda: hip.conv2d_bf16(x, hi, None, b, out, n, Hh, Ww, Cin, Cout, k, k, s, p, Cout), iters=5, warm=1)
        fl = 2 * n * Ho * Wo * Cout * (k * k * (3 if Cin == 4 else Cin))
        print(f"conv n={n} {Hh}x{Ww} {Cin}->{Cout} k{k} s{s}: " + "  ".join(f"{kk} {v:.0f} us ({fl / v / 1e6:.0f} TF/s)" for kk, v in r.items()))

if "lngemm" in which:
    for (M, N, K) in ((13056, 864, 256), (12288, 288, 256), (12288, 576, 256), (768, 864, 256)):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        Cm = torch.empty(M, N, device=dev)
        xn = torch.empty_like(A)
        hi, _ = split(W, False)
        t_g = timeit(lambda: hip.gemm_bf16(A, K, hi, None, K, b, None, 0, Cm, N, M, N, K, 0))
        t_l = timeit(lambda: hip.layernorm(A, K, None, None, xn, K, M, K, 1e-6))
        t_f = timeit(lambda: hip.ln_gemm_bf16(A, K, None, None, 1e-6, hi, None, K, b, None, 0, Cm, N, M, N, K, 0))
        print(f"M={M} N={N} K={K}: gemm {t_g:.1f} us, layernorm {t_l:.1f} us, ln_gemm {t_f:.1f} us")

if "block" in which:
    C, H, Ko = 256, 1024, 288
    def mk(n, k):
        hi = split(torch.randn(n, k, device=dev) / math.sqrt(k), False)[0]
        fr = torch.empty((n + 31) // 32 * 32 * k, device=dev, dtype=torch.int16)
        hip.pack_frag_bf16(hi, k, n, k, fr)
        return fr
    who, wh1, wh2 = mk(C, Ko), mk(H, C), mk(C, H)
    bo, b1, b2 = torch.randn(C, device=dev), torch.randn(H, device=dev), torch.randn(C, device=dev)
    for M in (12288, 768):
        x = torch.randn(M, C, device=dev)
        att = torch.randn(M, Ko, device=dev)
        for Ns in ([], [864], [576, 288]):
            nexts = [dict(w=mk(N, C), ldw=256, b=torch.randn(N, device=dev), N=N, y=torch.empty(M, N, device=dev), ldy=N, eps=1e-6) for N in Ns]
            t_f = timeit(lambda: hip.block_fused_bf16(x, C, att, Ko, Ko, who, Ko, bo, wh1, C, b1, wh2, H, b2, H, nexts, M, C))
            t_n = timeit(lambda: hip.block_fused_bf16(x, C, None, Ko, Ko, who, Ko, bo, wh1, C, b1, wh2, H, b2, H, [], M, C))
            extra = ""
            if M <= 2048:
                ws = torch.empty(5 * M * C, device=dev)
                t_s = timeit(lambda: hip.block_fused_bf16(x, C, att, Ko, Ko, who, Ko, bo, wh1, C, b1, wh2, H, b2, H, nexts, M, C, ws=ws))
                extra = f"   split path: {t_s:.1f} us"
            print(f"block_fused M={M} nexts={Ns}: {t_f:.1f} us   (MLP only: {t_n:.1f} us){extra}")

if "window" in which:
    # Secondary operator a7' (CorrBlock.corr_sample, spatracker/blocks.py:492-533) at its C3 shape: S=12 frames, N=1024 tracks,
    # C=128, 128x128 maps, 4 levels, r=4.  Algorithmic bytes per (frame, track, level) unit:
    # (2r+2)^2 texels x C x 4 B + target C x 4 B + coord 8 B read, (2r+1)^2 x 4 B written = 52 044 B.
    S, N, C, Hm, r, L = 12, 1024, 128, 128, 4, 4
    D = (2 * r + 1) ** 2
    unit = (2 * r + 2) ** 2 * C * 4 + C * 4 + 8 + D * 4
    maps = []
    f = torch.randn(S, Hm, Hm, C, device=dev)
    for lvl in range(L):
        maps.append(f.contiguous())
        f = torch.nn.functional.avg_pool2d(f.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    tg = torch.randn(S, N, C, device=dev)
    cd = torch.rand(S, N, 2, device=dev) * (Hm - 1)
    out = torch.zeros(S, N, L * D, device=dev)

    def run(levels=range(L)):
        for lvl in levels:
            h = maps[lvl].shape[1]
            hip.window_corr(maps[lvl], tg, cd, out, S, N, C, h, h, lvl, r, L * D, lvl * D)

    tot = timeit(run)
    per = [timeit(lambda l=l: run([l])) for l in range(L)]
    alg = S * N * unit
    print(f"window_corr C3 shape: 4 levels {tot:.1f} us = {L * alg / tot / 1e3:.0f} GB/s algorithmic ({L * alg / tot / 1e3 / 8000:.2f} of 8 TB/s); per level "
          + ", ".join(f"L{l} {t:.1f} us ({alg / t / 1e3:.0f} GB/s)" for l, t in enumerate(per)) + f"; {unit} B/unit, {alg / 1e6:.0f} MB per level launch")
