#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <name>   -> gpurun_out/<name>/...
# The profile set of one round: the unprofiled driver line, rocprofv3 kernel stats of the same command, HBM-traffic and MFMA-busy
# PMC passes (each in its own run, --kernel-trace only), and the secondary operator alone.
name=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$name
mkdir -p $out
cd $root
python3 bench.py --steps 20 --warmup 5 > $out/bench_bf16.json 2> $out/bench_bf16.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-corr-calibration > $out/bench_under_rocprof.json 2> $out/trace.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o $c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-corr-calibration > $out/$c.log 2>&1 || exit 1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/mfma -o m -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-corr-calibration > $out/mfma.log 2>&1 || exit 1
for mode in bf16 fp32; do
  python3 $root/tools/prof_window_corr.py $mode 20 > $out/window_corr_$mode.txt 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/wc_$mode -o w -- python3 $root/tools/prof_window_corr.py $mode 10 > $out/wc_$mode.log 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/wc_${mode}_$c -o $c -- python3 $root/tools/prof_window_corr.py $mode 4 > $out/wc_${mode}_$c.log 2>&1 || exit 1
  done
done
ls $out
