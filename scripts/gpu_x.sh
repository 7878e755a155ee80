#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 300 "$@" 2>/tmp/err.log | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()}, j.get('verified_vs_single_gpu'), j['roofline']['frac'])" || tail -5 /tmp/err.log; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 2>&1 | tail -3
for v in "" _m1; do
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip$v.so
echo "### lib$v"
run $B --verify
run $B --config c3 --verify
run $B --config c2 --verify
run $B --config c5 --steps 4 --verify
run $B --dtype u8 --verify
run $B --shard-sim 8
done
