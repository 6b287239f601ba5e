"""MFMA utilisation per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE
(tools/pmc_one.sh <dir> "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" bench.py --steps 1 --warmup 0 --no-cpu-baseline).
    python tools/mfma_busy_summary.py gpurun_out/<dir> profiles/r02_mfma_busy.json
SQ_VALU_MFMA_BUSY_CYCLES counts cycles with an MFMA executing, summed over the chip's 1024 SIMDs (32 per 32x32x16 bf16 MFMA);
GRBM_GUI_ACTIVE is the kernel's cycles summed over the 8 XCDs (MI355X_MICROARCH.md).  busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
per = defaultdict(lambda: defaultdict(float))
dur = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Dispatch_Id"])
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for (k, _), c in per.items():
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", k)
    n = re.sub(r"\(.*", "", n)
    a = agg[n]
    a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[1] += c.get("GRBM_GUI_ACTIVE", 0.0)
    a[2] += dur[(k, _)]
    a[3] += 1
rows = []
for n, (busy, gui, us, cnt) in agg.items():
    if busy <= 0 or gui <= 0:
        continue
    rows.append({"kernel": n, "launches": cnt, "mean_us_under_pmc": us / cnt, "mfma_busy_frac": busy / (gui / 8.0 * 1024.0),
                 "mfma_busy_cycles_per_launch": busy / cnt})
rows.sort(key=lambda r: -r["mfma_busy_cycles_per_launch"] * r["launches"])
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
           "note": "mfma_busy_frac = fraction of SIMD-cycles with an MFMA executing (1.0 = the dense bf16 MFMA peak at the clock held)", "kernels": rows},
          open(out, "w"), indent=1)
for r in rows[:14]:
    print(f'{r["kernel"][:60]:60s} n={r["launches"]:4d}  {r["mean_us_under_pmc"]:8.1f} us  MFMA busy {100 * r["mfma_busy_frac"]:5.1f} %')
