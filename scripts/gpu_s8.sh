#!/bin/bash
# short session: the tests that touch the post-processing rows (fillInv, weighted median) + bench.py --pp twice
TAG=${1:-s8}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_pp_ocv.py tests/test_gpu_fuzz.py tests/test_gpu_stripes.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for i in 1 2; do timeout 300 python bench.py --pp --steps 3 --warmup 1 --no-cpu-baseline --frame-loop 0 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print({k:v for k,v in j['pp'].items() if k!='note'})"; done | tee $O/pp.txt
