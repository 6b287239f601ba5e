"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh -> profiles/<name>.json for one kernel (substring filter).
    python tools/pmc_traffic_summary.py gpurun_out/<dir> corr_gather_dot <algorithmic bytes per launch> profiles/r02_corr_traffic.json
Corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes for gfx950: FETCH_SIZE counts the 128-B requests of
16-B-per-lane reads at 64 B -> x2; WRITE_SIZE is exact for 16-B-per-lane stores; unit KB = 1024 B."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, filt, alg, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
res = {}
name = None
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    per = defaultdict(float)
    for f in glob.glob(f"{root}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if filt in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
                name = r["Kernel_Name"]
    v = sorted(per.values())
    big = [x for x in v if x > 0.5 * v[-1]]  # the full-size launches (windows with every query)
    res[ctr] = big
rd = 2.0 * 1024 * sum(res["FETCH_SIZE"]) / len(res["FETCH_SIZE"])
wr = 1024 * sum(res["WRITE_SIZE"]) / len(res["WRITE_SIZE"])
j = {"kernel": name, "command": "tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, --kernel-trace only; "
     "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline)", "launches": len(res["FETCH_SIZE"]),
     "FETCH_SIZE_KB_mean": sum(res["FETCH_SIZE"]) / len(res["FETCH_SIZE"]), "WRITE_SIZE_KB_mean": sum(res["WRITE_SIZE"]) / len(res["WRITE_SIZE"]),
     "corrections": "gfx950: FETCH_SIZE counts 128-B requests of 16-B-per-lane reads at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); "
                    "WRITE_SIZE exact for 16-B-per-lane stores; unit KB = 1024 B",
     "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
     "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j, indent=1))
