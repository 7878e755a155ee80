#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 "$@" 2>/tmp/err.log | python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('  %.3f ms med %.3f'%(j['ms_per_step'], j['median_ms_per_step']), {k:v['avg_ms'] for k,v in j['kernels'].items()}, j.get('verified_vs_single_gpu'))" || tail -5 /tmp/err.log; }
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
echo "## two-phase"; run $B --verify; run $B --config c5 --steps 5 --verify
for S in 4 6 8; do for R in 2 4 8; do
echo "## S=$S R=$R"
PSM_PC_S=$S PSM_PC_S0=$R run $B --verify
done; done
PSM_PC_S=6 PSM_PC_S0=4 run $B --config c5 --steps 5 --verify
PSM_PC_S=4 PSM_PC_S0=4 run $B --config c5 --steps 5 --verify
PSM_PC_S=6 PSM_PC_S0=4 run $B --config c3 --verify
PSM_PC_S=6 PSM_PC_S0=4 run $B --dtype u8 --verify
