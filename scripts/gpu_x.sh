#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 "$@" 2>/tmp/err.log | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()})" || tail -5 /tmp/err.log; }
B="python bench.py --no-cpu-baseline --warmup 3"
for T in 1 2 3 4; do for S in 4 5 6; do
echo "## tail2=$T S=$S: c4 c5 c3 st8"
PSM_PC_TAIL=$T PSM_PC_S=$S run $B --steps 30
PSM_PC_TAIL=$T PSM_PC_S=$S run $B --steps 8 --config c5
PSM_PC_TAIL=$T PSM_PC_S=$S run $B --steps 30 --config c3
PSM_PC_TAIL=$T PSM_PC_S=$S run $B --steps 40 --shard-sim 8
done; done
