#!/bin/bash
# round-2 quick session: new tests first, then the whole gpu suite, then a bench line
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_pp_ocv.py tests/test_gpu_bench.py -m gpu -q -s -p no:cacheprovider --timeout=600 > $OUT/pytest_new.log 2>&1
grep -E "^\[wmf\]|^\[ocv-order\]|passed|failed|Error|error" $OUT/pytest_new.log | tail -40
echo "== whole suite"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err; cat $OUT/bench_c4.json; tail -2 $OUT/bench_c4.err
