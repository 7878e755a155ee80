#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()})"; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
run $B
run $B --flags 1048576
run $B
run $B --flags 1048576
run $B --config c3
run $B --config c3 --flags 1048576
