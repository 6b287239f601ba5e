python bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_c2.json 2>gpurun_out/r04_bench_c2.err
python bench.py --config c5shard --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_c5.json 2>gpurun_out/r04_bench_c5.err
MVT_OVERLAP=0 python bench.py --config c5shard --steps 3 --warmup 1 --no-cpu-baseline --no-corr-calibration > gpurun_out/r04_bench_c5_nooverlap.json 2>/dev/null
MVT_CONV_BIG_SHARED=0 python bench.py --config c5shard --steps 3 --warmup 1 --no-cpu-baseline --no-corr-calibration > gpurun_out/r04_bench_c5_bigoff.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-corr-calibration --clips-in-flight 2 > gpurun_out/r04_bench_twoclips.json 2>/dev/null
for f in c2 c5 c5_nooverlap c5_bigoff twoclips; do python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_$f.json"))
r=d["roofline"]; m=d["roofline_mfma"]
print("$f", round(d["ms_per_step"],2), "corr", round(r["frac"],3), r.get("frac_alone_warm"), r.get("frac_last_window"), "upd", round(m["updater"]["frac"],4), round(m["updater"]["ms_per_step"],1), "enc", round(m["encoder"]["frac"],4), round(m["encoder"]["ms_per_step"],1), d.get("throughput_two_clips",{}).get("ms_per_clip"))
PY
done
