"""Summarise rocprofv3 --pmc csv passes: per kernel (substring filter) mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

root, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(root + "/p*/*counter_collection.csv")):
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), d in per.items():
        for c, v in d.items():
            acc[k][c].append(v)
for k, d in acc.items():
    print(k[:110])
    for c, v in d.items():
        print(f"   {c:32s} {sum(v) / len(v):16.0f}   (n={len(v)})")
