#!/bin/bash
# weighted-median change check: its GPU tests, then the timing / trace of scripts/gpu_wm1.sh
TAG=${1:-wm2}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pp_ocv.py tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_stripes.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
for i in 1 2 3; do timeout 300 python bench.py --pp --steps 3 --warmup 1 --no-cpu-baseline --frame-loop 0 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(j['pp'])"; done | tee $O/pp3.txt
timeout 300 python scripts/dbg_wmf.py big > $O/wmf.txt 2> $O/wmf.err; cat $O/wmf.txt
bash scripts/gpu_wm1.sh $TAG
