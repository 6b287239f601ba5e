#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r03_v2; mkdir -p $out
cd $root && timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "window_corr" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for mode in bf16 fp32; do
  python3 $root/tools/prof_window_corr.py $mode 20 > $out/window_corr_$mode.txt 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/wc_$mode -o w -- python3 $root/tools/prof_window_corr.py $mode 10 > $out/wc_$mode.log 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/wc_${mode}_$c -o $c -- python3 $root/tools/prof_window_corr.py $mode 4 > $out/wc_${mode}_$c.log 2>&1 || exit 1
  done
done
grep -h window_corr $out/window_corr_*.txt
