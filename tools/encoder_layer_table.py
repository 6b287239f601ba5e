"""Per-layer TFLOP/s of the encoder convolutions from a rocprofv3 results.db of `bench.py` (C3: chunks of 24 images, 512x512).
    python tools/encoder_layer_table.py gpurun_out/<dir>/p_results.db > profiles/r02_encoder_layers.md
The bf16 conv kernels are identified by (template instantiation, grid size); conv2 and the four layer-1 convs share both and are
told apart by duration (conv2 has 6.5x the flops)."""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
n_img, H = 24, 512
if db.endswith(".csv"):  # rocprofv3 --kernel-trace --output-format csv
    import csv
    rows = [(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for r in sorted(csv.DictReader(open(db)), key=lambda r: int(r["Start_Timestamp"]))
            if "conv_rows_bf16" in r["Kernel_Name"] or "stem7x7" in r["Kernel_Name"] or "conv3x3_big_bf16" in r["Kernel_Name"]]
else:
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select name, grid_x, end-start from kernels where name like '%conv_rows_bf16%' or name like '%stem7x7%' or name like '%conv3x3_big_bf16%' order by start"))


def fl(ho, cout, cin, k):
    return 2.0 * n_img * ho * ho * cout * cin * k * k


LAYERS = {  # (template args, grid threads) -> [(label, flops)]
    ("stem", 1572864): [("conv1 7x7 s2 3->64 @256", fl(256, 64, 3, 7))],
    ("2, 2, 3, 1", 1572864): [("layer1 3x3 64->64 @256 (x4)", fl(256, 64, 64, 3)), ("conv2 3x3 416->256 @128", fl(128, 256, 416, 3))],
    ("big", 786432): [("conv2 3x3 416->256 @128, one 512-thread workgroup per CU (first block)", fl(128, 256, 416, 3))],
    ("1, 3, 3, 2", 786432): [("layer2.0.conv1 3x3 s2 64->96 @128", fl(128, 96, 64, 3))],
    ("1, 3, 3, 2 DS", 786432): [("layer2.0.conv1 3x3 s2 + downsample 1x1 s2 64->96 @128 (one launch)", fl(128, 96, 64, 3) + fl(128, 96, 64, 1))],
    ("1, 2, 3, 2 DS", 393216): [("layer3.0.conv1 3x3 s2 + downsample 96->128 @64 (one launch)", fl(64, 128, 96, 3) + fl(64, 128, 96, 1))],
    ("1, 2, 3, 2 DS", 98304): [("layer4.0.conv1 3x3 s2 + downsample 128->128 @32 (one launch)", fl(32, 128, 128, 3) + fl(32, 128, 128, 1))],
    ("2, 3, 3, 1", 393216): [("layer2 3x3 96->96 @128 (x3)", fl(128, 96, 96, 3))],
    ("2, 3, 1, 2", 393216): [("layer2 downsample 1x1 s2 64->96", fl(128, 96, 64, 1))],
    ("1, 2, 3, 2", 393216): [("layer3.0.conv1 3x3 s2 96->128 @64", fl(64, 128, 96, 3))],
    ("2, 2, 3, 1", 196608): [("layer3 3x3 128->128 @64 (x3)", fl(64, 128, 128, 3))],
    ("2, 2, 1, 2", 196608): [("layer3 downsample 1x1 s2 96->128", fl(64, 128, 96, 1))],
    ("1, 2, 3, 2", 98304): [("layer4.0.conv1 3x3 s2 128->128 @32", fl(32, 128, 128, 3))],
    ("2, 2, 3, 1", 49152): [("layer4 3x3 128->128 @32 (x3)", fl(32, 128, 128, 3))],
    ("2, 2, 1, 2", 49152): [("layer4 downsample 1x1 s2 128->128", fl(32, 128, 128, 1))],
    ("2, 2, 1, 1", 786432): [("conv3 1x1 256->128 @128", fl(128, 128, 256, 1))],
}
acc = defaultdict(list)
for name, grid, dur in rows:
    if "stem7x7" in name:
        key = "stem"
    elif "conv3x3_big" in name:
        key = "big"
    else:
        key = re.search(r"conv_rows_bf16<(\d, \d, \d, \d)", name).group(1)
        if re.search(r"conv_rows_bf16<\d, \d, \d, \d, \w+, \d, \w+, true>", name):
            key += " DS"
    cands = LAYERS.get((key, grid))
    if not cands:
        acc[(f"unmapped {key} grid {grid}", 0.0)].append(dur)
        continue
    if len(cands) == 2:  # layer1 vs conv2: by duration
        med = sorted(d for n2, g2, d in rows if g2 == grid and key in n2)[len(rows) // 200]
        lab = cands[1] if dur > 2.5 * min(d for n2, g2, d in rows if g2 == grid and key in n2) else cands[0]
    else:
        lab = cands[0]
    acc[lab].append(dur)
print("| layer (24 images of 512x512 per launch) | launches | mean us | GFLOP / launch | TFLOP/s | % of 2.5 PF |")
print("|---|---|---|---|---|---|")
tot_t = tot_f = 0.0
for (lab, f), ds in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    m = sum(ds) / len(ds) / 1e3
    tf = f / (m * 1e-6) / 1e12 if f else 0.0
    tot_t += sum(ds) / 1e3
    tot_f += f * len(ds)
    print(f"| {lab} | {len(ds)} | {m:.1f} | {f / 1e9:.0f} | {tf:.0f} | {100 * tf / 2500:.0f} |")
print(f"\nall convolution launches: {tot_t / 1e3:.2f} ms, {tot_f / 1e12:.2f} TFLOP -> {tot_f / tot_t / 1e6:.0f} TFLOP/s ({100 * tot_f / tot_t / 1e6 / 2500:.0f} % of the dense bf16 peak)")
