#!/bin/bash
# Round-3 session D: weighted median, lane-per-pixel evaluation - parity + timing
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pp_ocv.py tests/test_gpu_stripes.py -m gpu -q -x -p no:cacheprovider --timeout=900 2>&1 | tail -8
timeout 300 python scripts/dbg_wmf.py big > $OUT/wmf_timing.txt 2>&1; tail -12 $OUT/wmf_timing.txt
timeout 600 python bench.py --pp --no-cpu-baseline --steps 5 > $OUT/bench_pp.json 2> $OUT/bench.err; python -c "import json;j=json.load(open('$OUT/bench_pp.json'));print(j['pp'])"; tail -3 $OUT/bench.err
