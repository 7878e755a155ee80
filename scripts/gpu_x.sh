#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 "$@" 2>/tmp/err.log | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()})" || tail -5 /tmp/err.log; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
for S in 5 6 8 10 12 16; do
export PSM_PC_S=$S
echo "#### S=$S: c4 / c5 / c3 forced / c2 forced / shard-sim 8 disp forced / c3 stripe8"
run $B
run $B --config c5 --steps 4
run $B --config c3 --flags 1048576
run $B --config c2 --flags 1048576
run $B --shard-sim 8 --shard disp --flags 1048576
done
unset PSM_PC_S
echo "#### single-phase refs: c3 c2 ss8disp"
run $B --config c3
run $B --config c2
run $B --shard-sim 8 --shard disp
