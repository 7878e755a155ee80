#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 120 "$@" 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()}, j.get('verified_vs_single_gpu'))"; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider --timeout=300 2>&1 | tail -3
run $B --verify
PSM_PC_S=8 run $B --verify
PSM_PC_S=4 run $B --verify
run $B --flags 262144
run $B --config c5 --steps 3 --verify
run $B --config c3 --verify
run $B --config c2 --verify
