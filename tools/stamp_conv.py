"""Phase shares of the row-tile convolution from in-kernel s_memtime stamps (diagnostic build only):

    hipcc ... -DMVT_STAMPS -c mvtracker_amd/csrc/conv_rows.hip ; link as another library
    MVT_LIB=<that .so> python tools/stamp_conv.py [n H W Cin Cout]

Prints, for workgroups 0 and 2000 and every wave, the cycles between consecutive stamps (read the shares, not the length)."""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda"
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [24, 256, 256, 64, 64]
n, H, W, Cin, Cout = a
k, s, p = 3, 1, 1
lib = ctypes.CDLL(os.environ["MVT_LIB"])
K = k * k * Cin
ld = (K + 63) // 64 * 64
w = torch.zeros(Cout, ld, device=dev)
w[:, :K] = torch.randn(Cout, K, device=dev) / math.sqrt(K)
hi = torch.empty(Cout, ld, device=dev, dtype=torch.int16)
hip.split_bf16(w, hi, None, w.numel())
x = torch.randn(n, H, W, Cin, device=dev).to(torch.bfloat16)
out = torch.empty(n, H, W, Cout, device=dev, dtype=torch.bfloat16)
b = torch.randn(Cout, device=dev)
slots = hip.conv2d_stat_slots(H, W, Cin, k, k, s, p, False)
part = torch.empty(n * max(slots, 1) * Cout * 2, device=dev)
st = torch.rand(n, Cin, 2, device=dev) + 0.5
norm = os.environ.get("STAMP_NORM", "1") != "0"
run = lambda: hip.conv2d_bf16(x, hi, None, b, out, n, H, W, Cin, Cout, k, k, s, p, Cout, in_stats=st if norm else None, out_partial=part)
for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.mvt_debug_clear_conv_stamps() == 0
run()
torch.cuda.synchronize()
NWV = 8
buf = np.zeros(2 * NWV * 128, dtype=np.uint64)
assert lib.mvt_debug_read_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
buf = buf.reshape(2, NWV, 128).astype(np.int64)
names = {0: "start", 120: "loop end", 121: "epilogue", 122: "end"}
if os.environ.get("STAMP_RAW"):
    names = {}
for c in range(0 if os.environ.get('STAMP_RAW') else 14):
    names.update({1 + 8 * c: f"c{c} top", 2 + 8 * c: f"c{c} patch st"})
    for kh in range(3):
        names.update({3 + 8 * c + 2 * kh: f"c{c} kh{kh} bar", 4 + 8 * c + 2 * kh: f"c{c} kh{kh} mfma"})
for wg in range(2):
    if not (buf[wg] > 0).any():
        continue
    t0 = buf[wg][buf[wg] > 0].min()
    print(f"workgroup slot {wg}: cycles since the first stamp / delta, per wave")
    prev = {w_: None for w_ in range(NWV)}
    for i in [i for i in range(128) if (buf[wg, :, i] > 0).any()]:
        cells = []
        for w_ in range(NWV):
            v = buf[wg, w_, i]
            if v <= 0:
                cells.append("      -      ")
                continue
            d = v - prev[w_] if prev[w_] is not None else 0
            prev[w_] = v
            cells.append(f"{v - t0:6d}/{d:5d}")
        print(f"{i:3d} {names.get(i, ''):12s} " + " ".join(cells))
