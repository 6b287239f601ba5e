#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_window_corr_cache.sh <name>  -> gpurun_out/<name>/...
# What bounds the secondary operator (mvt_window_corr_levels, bf16 maps, C3 shape): texture-address / L1 / L2 counters of the
# launch, one --pmc pass per counter group (each with --kernel-trace only), summarised by tools/pmc_summary.py.
name=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 -L > $out/counters.txt 2>&1
grep -oE "\b(TCP|TA|TD|TCC)_[A-Za-z0-9_]+" $out/counters.txt | sort -u > $out/counter_names.txt
i=0
# (at most two counters of a block per pass: more "exceeds the capabilities of the hardware" and the profiler aborts -- every pass under
#  its own timeout, a failed pass is recorded and skipped)
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ok=""
  for c in $set; do if grep -qx "$c" $out/counter_names.txt || [[ $c == SQ_* || $c == GRBM_* ]]; then ok="$ok $c"; fi; done
  [ -z "$ok" ] && continue
  echo "pass $i:$ok" >> $out/passes.txt
  timeout -k 5 90 rocprofv3 --pmc $ok --kernel-trace --output-format csv -d $out/p$i -o p$i -- python3 $root/tools/prof_window_corr.py bf16 4 > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/passes.txt
done
python3 $root/tools/pmc_summary.py $out window_corr_levels > $out/summary.txt 2>&1
cat $out/passes.txt $out/summary.txt
