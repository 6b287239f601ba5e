"""Random shapes through the bf16 convolution entry (mvt_conv2d_bf16: row tiles, the wide 512-thread kernel, stride 2, 1x1,
normalise-on-load, fused statistics) against an fp64 convolution of the same bf16 operands.  A robustness sweep for the GPU box:

    python tests/checks/fuzz_conv.py [n_configs] [first_seed]
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvtracker_amd import hip  # noqa: E402

DEV = "cuda"
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
for kcfg in range(n_cfg):
    rng = np.random.default_rng(500 + seed0 + kcfg)
    n = int(rng.integers(1, 4))
    H, W = int(rng.integers(5, 75)), int(rng.integers(5, 140))
    Cin = int(rng.choice([32, 64, 96, 128, 416]))
    Cout = int(rng.choice([32, 64, 96, 128, 256]))
    k, s = [(3, 1), (3, 1), (3, 2), (1, 1), (1, 2)][int(rng.integers(5))]
    p = 1 if k == 3 else 0
    norm = bool(rng.integers(2)) and k == 3 and s == 1
    big = os.environ.get("MVT_CONV_BIG", "1") != "0" and k == 3 and s == 1 and not norm and Cout % 256 == 0
    tag = f"cfg {kcfg}: n={n} {H}x{W} {Cin}->{Cout} k{k}s{s} norm={norm}{' (wide kernel)' if big else ''}"
    try:
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = (torch.randn(n, H, W, Cin, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
        w = (torch.randn(Cout, k, k, Cin, generator=g) / math.sqrt(Cin * k * k))
        b = torch.randn(Cout, generator=g)
        K = k * k * Cin
        wp = torch.zeros(Cout, (K + 63) // 64 * 64)
        wp[:, :K] = w.reshape(Cout, K)
        hi = torch.empty(wp.shape, device=DEV, dtype=torch.int16)
        hip.split_bf16(wp.to(DEV), hi, None, wp.numel())
        wb = w.to(torch.bfloat16).double()  # (split_bf16 rounds to nearest even, as .to(bfloat16) does)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        st = None
        xr = x.double()
        if norm:
            st = torch.stack([torch.randn(n, Cin, generator=g) * 0.2, torch.rand(n, Cin, generator=g) + 0.5], -1)
            xn = torch.relu((x.float() - st[:, None, None, :, 0]) * st[:, None, None, :, 1]).to(torch.bfloat16)  # the loader's two roundings
            xr = xn.double()
        ref = F.conv2d(xr.permute(0, 3, 1, 2), wb.permute(0, 3, 1, 2), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
        slots = hip.conv2d_stat_slots(H, W, Cin, k, k, s, p, False)
        out = torch.full((n, Ho, Wo, Cout), float("nan"), device=DEV).to(torch.bfloat16)
        part = torch.full((n * max(slots, 1) * Cout * 2,), float("nan"), device=DEV)
        hip.conv2d_bf16(x.to(DEV), hi, None, b.to(DEV), out, n, H, W, Cin, Cout, k, k, s, p, Cout, in_stats=None if st is None else st.to(DEV),
                        out_partial=part if slots else None)
        torch.cuda.synchronize()
        o = out.double().cpu()
        assert bool(torch.isfinite(o).all()), "non-finite / unwritten output"
        err = ((o - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert err < 6e-3, f"output error {err:.2e}"  # bf16 rounding of the output (2^-8 relative) + fp32 accumulation
        if slots:
            stat = torch.empty(n, Cout, 2, device=DEV)
            hip.instnorm_finish_slots(part, slots, stat, n, Ho * Wo, Cout)
            torch.cuda.synchronize()
            y = ref.reshape(n, Ho * Wo, Cout)
            mean, var = y.mean(1), y.var(1, unbiased=False)
            em = (stat[..., 0].double().cpu() - mean).abs().max().item()
            er = ((stat[..., 1].double().cpu() * torch.sqrt(var + 1e-5)) - 1).abs().max().item()
            assert em < 2e-3 and er < 2e-3, f"statistics: mean {em:.2e} rstd {er:.2e}"  # (of the UNROUNDED fp32 sums vs fp64)
        print(f"ok   {tag}: {err:.2e}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{n_cfg - fails} / {n_cfg} configurations passed")
sys.exit(1 if fails else 0)
