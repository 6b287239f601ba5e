#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_cache.sh <name> <kernel substring> <script.py> [args...]  -> gpurun_out/<name>/...
# Texture-address / L1 / L2 counters of one kernel of a small target script (e.g. tools/prof_conv.py), one --pmc pass per counter group
# (each with --kernel-trace only, at most two counters of a block per pass), summarised by tools/pmc_summary.py.
name=$1; filt=$2; shift 2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set" >> $out/passes.txt
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o p$i -- python3 $root/"$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/passes.txt
done
python3 $root/tools/pmc_summary.py $out "$filt" > $out/summary.txt 2>&1
cat $out/passes.txt $out/summary.txt
