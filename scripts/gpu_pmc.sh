#!/bin/bash
# PMC passes (SQ issue / wait / LDS counters) of the fused kernels: default (k_cvf_q2) and PSM_FLAGS variants
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for var in "q2:1" "pc:16384"; do
  v=${var%%:*}; f=${var#*:}
  for pass in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sqb:SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8" "lds:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
    n=${pass%%:*}; c=${pass#*:}
    PSM_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/${v}_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 0 > $OUT/${v}_$n.log 2>&1 || echo "pmc pass $v $n failed"
    fdb=$(find $OUT/${v}_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/${v}_$n.summary.txt 2>&1
  done
done
find $OUT -name "*.db" -delete
grep -h -A12 "counters: .*k_cvf" $OUT/*.summary.txt | head -150
