#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 120 "$@" 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()}, j.get('verified_vs_single_gpu'))"; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
for v in "" _occ4; do
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip$v.so
echo "### lib$v"
run $B --verify
run $B --flags 2097152
done
