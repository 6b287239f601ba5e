#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_stats.sh <name> [ENV=VAL ...]  -> gpurun_out/<name>_kernel_stats.csv
# rocprofv3 kernel statistics of a short bench run (5 steps); the environment assignments select tuning switches.
name=$1; shift
for kv in "$@"; do export "$kv"; done
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-corr-calibration > $out/bench.json 2> $out/trace.err || exit 1
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cp $f $root/gpurun_out/${name}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:16]:
    print(r['Name'][:78].ljust(78), r['Calls'].rjust(5), '%9.1f' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
