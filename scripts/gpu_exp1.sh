#!/bin/bash
# round 5, session 1: frames-in-flight experiment + PMC passes (HBM counters) for BASELINE configs[2] (c3), c2 and c5
TAG=${1:-r5a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
GFX=$(rocminfo 2>/dev/null | grep -m1 -oE "gfx9[0-9a-f]+")
CPU=$(grep -m1 "model name" /proc/cpuinfo | sed 's/.*: //')
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ): $GFX (MI355X), host $CPU, $(nproc) cores, box $(hostname)" > $OUT/device.txt
echo "== frames in flight"
timeout 600 python scripts/exp_inflight.py > $OUT/inflight.txt 2>&1; cat $OUT/inflight.txt
echo "== PMC passes c3 / c2 / c5 (own runs, kernel trace only)"
cd /tmp
for cfg in c3 c2 c5; do
  for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr:WRITE_SIZE TCC_EA0_WRREQ_sum" "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    n=${pass%%:*}; c=${pass#*:}
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${cfg}_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py $cfg > $OUT/pmc_${cfg}_$n.log 2>&1 || echo "pmc pass $cfg $n failed"
    fdb=$(find $OUT/pmc_${cfg}_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_${cfg}_$n.summary.txt 2>&1
  done
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*.db" -delete
ls $OUT
echo "== done"
