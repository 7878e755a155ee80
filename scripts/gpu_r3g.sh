#!/bin/bash
# Round-3 session G: PMC comparison of the key phase, f32 vs 8-bit
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for dt in f32 u8; do
for pass in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sqb:SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "tcc:TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "wr:WRITE_SIZE TCC_EA0_WRREQ_sum"; do
  n=${pass%%:*}; c=${pass#*:}
  PSM_DTYPE=$dt timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${dt}_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 > $OUT/pmc_${dt}_$n.log 2>&1 || echo "pmc pass $dt $n failed"
  fdb=$(find $OUT/pmc_${dt}_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_${dt}_$n.summary.txt 2>&1
done; done
find $OUT -name "*.db" -delete
grep -h -A9 "counters: void psm::k_cvf_pc<false, 3, 2" $OUT/pmc_*.summary.txt | cut -c1-120
