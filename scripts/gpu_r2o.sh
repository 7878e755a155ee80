#!/bin/bash
TAG=${1:-r02o}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "bench or dist or shard or flags" > $OUT/pytest_sel.log 2>&1; tail -4 $OUT/pytest_sel.log
cd /tmp
for g in 8 1; do
  extra=""; [ $g -gt 1 ] && extra="--shard-sim $g"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$g -o t -- python $GRAFT_REPO_ROOT/bench.py $extra --no-cpu-baseline --steps 20 --warmup 3 > $OUT/trace_$g.log 2>&1
  f=$(find $OUT/trace_$g -name "*kernel_trace.csv" | head -1)
  echo "== shard-sim $g"; python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f 60 | head -30
done
find $OUT -name "*.csv" -size +5M -delete
