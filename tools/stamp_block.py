"""Phase shares of the fused updater block from in-kernel s_memtime stamps (diagnostic build only):

    hipcc ... -DMVT_STAMPS -c mvtracker_amd/csrc/block_fused.hip ; link as another library ; MVT_LIB=<that .so> python tools/stamp_block.py

Prints, for workgroups 0 and 100 and every wave, the cycles between consecutive stamps.  Read the SHARES, not the length (the
stamps' fences forbid overlaps the shipped kernel has)."""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import hip  # noqa: E402

dev = "cuda"
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"   # plain | time | p2v
n_tracks, S = 1024, 12
C, H, Ko = 256, 1024, 288
lib = ctypes.CDLL(os.environ["MVT_LIB"]) if (os.environ.get("MVT_LIB") and not os.environ.get("TIME_ONLY")) else None  # TIME_ONLY=1: time another build


def mk(n, k):
    w = torch.randn(n, k, device=dev) / math.sqrt(k)
    hi = torch.empty(n, k, device=dev, dtype=torch.int16)
    hip.split_bf16(w, hi, None, n * k)
    fr = torch.empty((n + 31) // 32 * 32 * k, device=dev, dtype=torch.int16)
    hip.pack_frag_bf16(hi, k, n, k, fr)
    return fr


who, wh1, wh2 = mk(C, Ko), mk(H, C), mk(C, H)
bo, b1, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
Mp, Mv = n_tracks * S, 64 * S
M = Mp + Mv if kind == "time" else Mp
x = torch.randn(M, C, device=dev)
bf = lambda *sh: (torch.randn(*sh, device=dev)).to(torch.bfloat16)


def nxt(N, rows=(0, 0), affine=False):
    d = dict(w=mk(N, C), ldw=256, b=torch.randn(N, device=dev) * 0.1, N=N, y=torch.empty(Mp + Mv, N, device=dev, dtype=torch.bfloat16), ldy=N,
             eps=1e-6, rows=rows)
    if affine:
        d["lnw"], d["lnb"] = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    return d


if kind == "plain":
    att = bf(M, Ko)
    nexts = [nxt(864)]
    run = lambda: hip.block_fused_bf16(x, C, att, Ko, Ko, who, Ko, bo, wh1, C, b1, wh2, H, b2, H, nexts, M, C)
elif kind == "time":
    qkv = bf(M, 864)
    nexts = [nxt(576, (0, Mp), True), nxt(288, (0, Mp)), nxt(288, (Mp, M))]
    run = lambda: hip.attn_block_fused_bf16(x, C, hip.ATTN_TIME, S, qkv, 864, qkv[:, 288:], qkv[:, 576:], 864, S, who, bo, wh1, b1, wh2, b2, H,
                                            nexts, M, C)
else:
    q = bf(Mp, 288)
    kv = bf(Mv, 864)
    nexts = [nxt(864)]
    run = lambda: hip.attn_block_fused_bf16(x, C, hip.ATTN_FRAME, S, q, 288, kv[:, 288:], kv[:, 576:], 864, 64, who, bo, wh1, b1, wh2, b2, H,
                                            nexts, Mp, C)
for _ in range(3):
    run()
torch.cuda.synchronize()
if lib is None:  # shipped library: time only
    ts = []
    for rep in range(int(os.environ.get("REPS", "6"))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts = ts[1:]
    print(kind, f"us per launch: min {min(ts):.1f} median {sorted(ts)[len(ts) // 2]:.1f} |", " ".join(f"{t:.1f}" for t in ts[:8]))
    sys.exit(0)
assert lib.mvt_debug_clear_stamps() == 0
run()
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 64, dtype=np.uint64)
assert lib.mvt_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
buf = buf.reshape(2, 8, 64).astype(np.int64)
print("kind", kind)
names = {0: "start", 1: "pre-attn", 2: "tile staged", 3: "barrier", 4: "out-proj", 5: "x added", 6: "LN1", 30: "mlp end", 31: "store x", 63: "end"}
names.update({49: "attn loads issued", 50: "unit 0", 51: "unit 1", 52: "unit 2", 53: "unit 3"})
for c in range(4):
    names.update({8 + 5 * c: f"c{c} begin", 9 + 5 * c: f"c{c} fc1", 10 + 5 * c: f"c{c} gelu+st", 11 + 5 * c: f"c{c} barrier", 12 + 5 * c: f"c{c} fc2"})
for q in range(3):
    names.update({32 + 8 * q: f"p{q} begin", 33 + 8 * q: f"p{q} LN", 39 + 8 * q: f"p{q} end"})
    for b in range(5):
        names[34 + 8 * q + b] = f"p{q} blk{b}"
for wg in range(2):
    if not (buf[wg] > 0).any():
        continue
    t0 = buf[wg][buf[wg] > 0].min()
    print(f"workgroup {'0' if wg == 0 else '100'}: cycles since the first stamp / delta, per wave")
    idx = [i for i in range(64) if (buf[wg, :, i] > 0).any()]
    prev = {w: None for w in range(8)}
    for i in idx:
        cells = []
        for w in range(8):
            v = buf[wg, w, i]
            if v <= 0:
                cells.append("      -      ")
                continue
            d = v - prev[w] if prev[w] is not None else 0
            prev[w] = v
            cells.append(f"{v - t0:6d}/{d:5d}")
        print(f"{i:2d} {names.get(i, ''):12s} " + " ".join(cells))
