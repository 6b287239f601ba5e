#!/bin/bash
# HBM traffic of the kernels of one bench step, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (each with --kernel-trace only).  Run on the GPU box from the repo root: tools/pmc_traffic.sh <outdir>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-corr-calibration > $out/$c.log 2>&1 || exit 1
done
ls $out/*
