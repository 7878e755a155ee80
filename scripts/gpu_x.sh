#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 "$@" 2>/tmp/err.log | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()})" || tail -5 /tmp/err.log; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
for rep in 1 2; do
for sl in 96 128; do
echo "## slots $sl"
PSM_PC_SLOTS=$sl run $B
PSM_PC_SLOTS=$sl run $B --config c5 --steps 5
PSM_PC_SLOTS=$sl run $B --config c3
PSM_PC_SLOTS=$sl run $B --shard-sim 8
done; done
