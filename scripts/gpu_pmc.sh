#!/bin/bash
# rocprofv3 PMC passes (own runs, kernel-trace only) for the hot kernels.  bash scripts/gpu_pmc.sh <tag> [config] [seg_rows]
TAG=${1:-pmc}; CFG=${2:-c4}; SR=${3:-135}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py $CFG $SR > $OUT/$n.log 2>&1 || echo "pass $n failed"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
run sq3 SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_ADD_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run sq4 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_BRANCH
cd $GRAFT_REPO_ROOT
for n in sq1 sq2 sq3 sq4 tcc1 tcc2 tcc3 tcp1; do
  f=$(find $OUT/$n -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/rocpd_summary.py $f > $OUT/$n.summary.txt 2>&1
  tail -3 $OUT/$n.log | cut -c1-200
done
find $OUT -name "*.db" -size +30M -delete
echo pmc done
