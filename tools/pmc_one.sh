#!/bin/bash
# usage: tools/pmc_one.sh <outdir> "<counters>" <script> [args]   -- one rocprofv3 --pmc pass (kernel-trace only)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ctr="$1"; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/p1 -o p1 -- python3 $GRAFT_REPO_ROOT/"$@" > $out/p1.log 2>&1
