"""One encoder convolution shape, a few launches: target for rocprofv3 --pmc passes.

    python3 tools/prof_conv.py [n H W Cin Cout k stride pad] [bf16_tensors]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda"
a = [int(v) for v in sys.argv[1:9]] if len(sys.argv) >= 9 else [16, 256, 256, 64, 64, 3, 1, 1]
n, H, W, Cin, Cout, k, s, p = a
bf = len(sys.argv) > 9 and sys.argv[9] == "1"
norm = not (len(sys.argv) > 10 and sys.argv[10] == "0")  # 10th argument 0: no normalise-on-load (conv2 reads materialised activations)
stem = Cin == 4
K = k * 32 if stem else k * k * Cin
ld = (K + 63) // 64 * 64
w = torch.zeros(Cout, ld, device=dev)
w[:, :K] = torch.randn(Cout, K, device=dev) / math.sqrt(K)
hi = torch.empty(Cout, ld, device=dev, dtype=torch.int16)
hip.split_bf16(w, hi, None, w.numel())
dt = torch.bfloat16 if bf else torch.float32
x = torch.randn(n, H, W, Cin, device=dev).to(torch.float32 if stem else dt)
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
out = torch.empty(n, Ho, Wo, Cout, device=dev, dtype=dt)
b = torch.randn(Cout, device=dev)
slots = hip.conv2d_stat_slots(H, W, Cin, k, k, s, p, False)
part = torch.empty(n * max(slots, 1) * Cout * 2, device=dev)
st = torch.rand(n, Cin, 2, device=dev) + 0.5
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
NIT = int(os.environ.get("PROF_CONV_ITERS", "6"))  # (6: a small target for PMC passes; more for timing A/B runs)
for it in range(NIT):
    if it == 1:
        ev0.record()
    hip.conv2d_bf16(x, hi, None, b, out, n, H, W, Cin, Cout, k, k, s, p, Cout, in_stats=st if (k == 3 and s == 1 and norm) else None,
                    out_partial=part if slots else None)
ev1.record()
torch.cuda.synchronize()
t = ev0.elapsed_time(ev1) / (NIT - 1) * 1e3
fl = 2.0 * n * Ho * Wo * Cout * (147 if stem else K)
print(f"conv n={n} {H}x{W} {Cin}->{Cout} k{k}s{s} bf16_tensors={bf}: {t:.1f} us  {fl / t / 1e6:.0f} TFLOP/s")
