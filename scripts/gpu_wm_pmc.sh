#!/bin/bash
# PMC passes over the weighted median of the 1080p bench pair (what binds k_wm_eval?  profiles/r06/exp_wgt_median.txt).
# SQ and TCC counters only: passes with TA_* / TCP_*_sum counters hang rocprofv3 on this pool until their timeout.
TAG=${1:-wmpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for pass in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "sqb:SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
            "lds:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
            "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=${pass%%:*}; c=${pass#*:}
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$n -o $n -- python $GRAFT_REPO_ROOT/bench.py --pp --steps 1 --warmup 1 --no-cpu-baseline --frame-loop 0 > $OUT/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
  fdb=$(find $OUT/pmc_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_$n.summary.txt 2>&1
  grep -A9 "counters: void psm::k_wm_eval<true>" $OUT/pmc_$n.summary.txt | head -10
done
