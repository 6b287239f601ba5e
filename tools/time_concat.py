"""The encoder's concat + resize launch alone (24 images of 512x512: four maps -> 128x128x416, bf16): us per launch.
    python tools/time_concat.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dims = [(256, 256, 64), (128, 128, 96), (64, 64, 128), (32, 32, 128)]
srcs = [torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16) for h, w, c in dims]
dst = torch.empty(n, 128, 128, 416, device="cuda", dtype=torch.bfloat16)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(41):
    if it == 1:
        ev0.record()
    hip.concat_resize_bilinear_ac(srcs, dims, dst, n, 128, 128, 416)
ev1.record()
torch.cuda.synchronize()
nbytes = dst.numel() * 2 + sum(t.numel() * 2 for t in srcs)
t = ev0.elapsed_time(ev1) / 40 * 1e3
print(f"concat_resize n={n}: {t:.1f} us, {nbytes / t / 1e6:.2f} TB/s of unique bytes; checksum {dst.float().sum().item():.6e}")
