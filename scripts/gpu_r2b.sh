#!/bin/bash
TAG=${1:-r02b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
timeout 900 python -m pytest tests/test_gpu_pp_ocv.py -m gpu -q -s -p no:cacheprovider --timeout=600 -k "wgt or create" > $OUT/pytest_new.log 2>&1
grep -E "\[wmf\]|passed|failed|Error|error" $OUT/pytest_new.log | tail -40
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
