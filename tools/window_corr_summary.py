"""Summary of the secondary operator's profile passes (tools/profile_round.sh): kernel duration from the rocprofv3 kernel trace,
HBM-side bytes from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction: FETCH_SIZE x 2 for 16-B-per-lane reads), against the
algorithmic bytes of SURVEY section 8d and the bytes the launch can touch at most once (maps + targets + coordinates + output).
    python tools/window_corr_summary.py gpurun_out/<round dir> profiles/r03_window_corr.json"""
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
S, N, C, Hm, r, L = 12, 1024, 128, 128, 4, 4
D = (2 * r + 1) ** 2
res = {"shape": f"S={S} frames x N={N} tracks, {Hm}x{Hm} maps of C={C}, r={r}, {L} levels, ONE launch for all levels (mvt_window_corr_levels)",
       "corrections": "gfx950: FETCH_SIZE counts 128-B requests of 16-B-per-lane reads at 64 B -> x2; unit KB = 1024 B (MI355X_MICROARCH.md)"}
for mode, esz in (("bf16", 2), ("fp32", 4)):
    durs = []
    for f in glob.glob(f"{root}/wc_{mode}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "window_corr_levels" in row["Kernel_Name"]:
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    ctr = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = {}
        for f in glob.glob(f"{root}/wc_{mode}_{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "window_corr_levels" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        ctr[c] = sum(per.values()) / max(len(per), 1)
    unit = (2 * r + 2) ** 2 * C * esz + C * 4 + 8 + D * 4
    alg = S * N * L * unit
    maps = [S * (Hm >> l) ** 2 * C * esz for l in range(L)]
    uniq = sum(maps) + S * N * C * 4 + S * N * 8 + S * N * L * D * 4
    us = sum(durs[3:]) / len(durs[3:])
    hbm = 2.0 * 1024 * ctr["FETCH_SIZE"] + 1024 * ctr["WRITE_SIZE"]
    res[mode] = {"launches_timed": len(durs) - 3, "avg_launch_us": us, "min_launch_us": min(durs),
                 "bytes_per_unit_and_level": unit, "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / us / 1e3,
                 "algorithmic_over_hbm_peak": alg / us / 1e3 / 8000.0,
                 "unique_bytes_per_launch": uniq, "map_bytes_per_level": maps,
                 "FETCH_SIZE_KB": ctr["FETCH_SIZE"], "WRITE_SIZE_KB": ctr["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                 "hbm_GBps": hbm / us / 1e3, "hbm_over_algorithmic": hbm / alg, "hbm_over_unique": hbm / uniq}
res["reading"] = ("The algorithmic figure counts every texel of every window: 100 texels x 1024 tracks per frame against 16 384 / 4 096 / 1 024 / "
                  "256 texels in the level's map, i.e. each texel is read 6 / 25 / 100 / 400 times per frame.  The HBM-side counters see "
                  "hbm_over_algorithmic of those bytes: the re-reads of ALL levels are served by L2 (a frame's level-0 map is 4 MB in bf16 -- one "
                  "XCD's L2 -- and 8 MB in fp32) and the Infinity Cache; the kernel is bound by L2 / TA request throughput (one 1-KiB wave "
                  "request per 4 (bf16) or 2 (fp32) texels), not by HBM.  'algorithmic_over_hbm_peak' above 1 is therefore not an HBM "
                  "fraction; the HBM-side rate is hbm_GBps.")
json.dump(res, open(out, "w"), indent=1)
for m in ("bf16", "fp32"):
    x = res[m]
    print(f"{m}: {x['avg_launch_us']:.1f} us, algorithmic {x['algorithmic_bytes_per_launch'] / 1e6:.0f} MB = {x['algorithmic_GBps']:.0f} GB/s, "
          f"HBM side {x['hbm_bytes_per_launch'] / 1e6:.0f} MB = {x['hbm_GBps']:.0f} GB/s ({x['hbm_over_algorithmic']:.3f} of algorithmic, "
          f"{x['hbm_over_unique']:.2f} of the unique bytes {x['unique_bytes_per_launch'] / 1e6:.0f} MB)")
