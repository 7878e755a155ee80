#!/bin/bash
# Round-3 session B: the GPU suite on the cleaned-up library (retired variants, split C-ABI layer, new entry points).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 --durations=15 > $OUT/pytest_gpu.log 2>&1
tail -40 $OUT/pytest_gpu.log
free -g | head -2; nproc
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B --verify > $OUT/bench_c4.json 2> $OUT/bench.err; python -c "import json;j=json.load(open('$OUT/bench_c4.json'));print(j['ms_per_step'],j['kernels'],j.get('verified_vs_single_gpu'))"; tail -3 $OUT/bench.err
