#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 "$@" 2>/tmp/err.log | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()}, j.get('verified_vs_single_gpu'))" || tail -5 /tmp/err.log; }
B="python bench.py --no-cpu-baseline --warmup 3 --steps 30"
for rep in 1 2; do for v in "" _mix; do
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip$v.so
echo "### lib$v"
run $B --verify
run $B --config c3
done; done
