"""The secondary operator (CorrBlock.corr_sample, bilinear (2r+1)^2 windows) at the C3 shape -- S = 12 frames, N = 1024 tracks, 128x128
maps of C = 128 channels, r = 4, 4 levels -- as ONE launch for all levels (mvt_window_corr_levels), bf16 maps (the dtype of config C3)
and fp32 maps: target for rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, and HIP-event timing when run alone.

    python3 tools/prof_window_corr.py [bf16|fp32] [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
S, N, C, Hm, r, L = 12, 1024, 128, 128, 4, 4
gen = torch.Generator().manual_seed(5)
dt = torch.bfloat16 if mode == "bf16" else torch.float32
pyr = [torch.randn(S, Hm >> l, Hm >> l, C, generator=gen).to(dev).to(dt).contiguous() for l in range(L)]
tg = torch.randn(S, N, C, generator=gen).to(dev)
cd = (torch.rand(S, N, 2, generator=gen) * (Hm - 1)).to(dev)
D = (2 * r + 1) ** 2
out = torch.zeros(S, N, L * D, device=dev)
for _ in range(3):
    hip.window_corr_levels(pyr, tg, cd, out, S, N, C, r, L * D)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    hip.window_corr_levels(pyr, tg, cd, out, S, N, C, r, L * D)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
esz = 2 if mode == "bf16" else 4
alg_unit = (2 * r + 2) ** 2 * C * esz + C * 4 + 8 + D * 4   # texels + target + coord read, window written
alg = S * N * L * alg_unit
uniq = sum(p.numel() * esz for p in pyr) + tg.numel() * 4 + cd.numel() * 4 + out.numel() * 4   # every byte touched once
print(f"window_corr_levels {mode}: {us:.1f} us per launch (4 levels); algorithmic {alg / 1e6:.1f} MB ({alg_unit} B per unit and level) -> "
      f"{alg / us / 1e3:.0f} GB/s of algorithmic bytes; unique bytes (maps + targets + coords + output) {uniq / 1e6:.1f} MB -> {uniq / us / 1e3:.0f} GB/s")
