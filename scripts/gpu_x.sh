#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.3f ms'%j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items() if k in ('cvf_fused','wta')})"; }
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
for seg in 135 90 68; do run $B --seg-rows $seg; run $B --seg-rows $seg --flags 524288; done
for dc in 1 2 4 8; do run env PSM_PC_DC=$dc $B --flags 524288 --seg-rows 135; done
