#!/bin/bash
# rocprofv3 kernel-trace of the FGF pipeline.  bash scripts/gpu_fgf_prof.sh <tag> [rates...]
TAG=${1:-fgfprof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
cd /tmp && export TMPDIR=/tmp
for s in ${@:-4}; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/s$s -o fgf_s$s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --fgf $s > $OUT/bench_s$s.log 2>&1
  f=$(find $OUT/s$s -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/fgf_s${s}_kernel_stats.csv && head -12 $f | cut -c1-220
  find $OUT/s$s -name "*kernel_trace.csv" -delete
done
