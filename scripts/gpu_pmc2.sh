#!/bin/bash
# two SQ counter passes over the fused filter (own runs, kernel-trace only).  bash scripts/gpu_pmc2.sh <tag> [flags]
TAG=${1:-pmc2}; FLAGS=${2:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  PSM_FLAGS=$FLAGS timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 0 > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
run sqa SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
run sqb SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
for n in sqa sqb; do
  f=$(find $OUT/$n -name "*.db" | head -1)
  [ -n "$f" ] && python scripts/rocpd_summary.py $f > $OUT/$n.summary.txt 2>&1
  grep -A10 "counters: void psm::.*k_cvf_pc" $OUT/$n.summary.txt | head -48
done
find $OUT -name "*.db" -size +30M -delete
