"""Time the fused updater block alone (HIP events, median of 5 batches of 20 launches).  MVT_LIB selects the library build.

    python tools/time_block.py [M] [N_next]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
N = int(sys.argv[2]) if len(sys.argv) > 2 else 768
C, H, Ko = 256, 1024, 288


def mk(n, k):
    w = torch.randn(n, k, device=dev) / math.sqrt(k)
    hi = torch.empty(n, k, device=dev, dtype=torch.int16)
    hip.split_bf16(w, hi, None, n * k)
    fr = torch.empty((n + 31) // 32 * 32 * k, device=dev, dtype=torch.int16)
    hip.pack_frag_bf16(hi, k, n, k, fr)
    return fr


who, wh1, wh2 = mk(C, Ko), mk(H, C), mk(C, H)
bo, b1, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
x0 = torch.randn(M, C, device=dev)
att = torch.randn(M, Ko, device=dev)
nexts = [dict(w=mk(N, C), ldw=256, b=torch.randn(N, device=dev), N=N, y=torch.empty(M, N, device=dev), ldy=N, eps=1e-6)] if N else []
ts = []
for rep in range(6):
    x = x0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.block_fused_bf16(x, C, att, Ko, Ko, who, Ko, bo, wh1, C, b1, wh2, H, b2, H, nexts, M, C)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(os.environ.get("MVT_LIB", "default"), "M", M, "N", N, "us per launch:", " ".join(f"{t:.1f}" for t in ts[1:]))
