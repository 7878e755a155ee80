#!/bin/bash
# PMC passes over the weighted median of the 1080p bench pair (what binds k_wm_eval?)
TAG=${1:-wmpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "ta:TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
            "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "utc:TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum" \
            "vm:SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  n=${pass%%:*}; c=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/dbg_wmf.py hd20 > $OUT/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
  fdb=$(find $OUT/pmc_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_$n.summary.txt 2>&1
  grep -A8 "counters: void psm::k_wm_eval<true>" $OUT/pmc_$n.summary.txt | head -9
done
grep "k_wm_eval<true>" $OUT/pmc_ta.summary.txt | head -1 | cut -c1-160
