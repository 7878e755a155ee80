#!/bin/bash
# weighted-median tail study: per-sweep counters (experiment build) + kernel trace of bench.py --pp
TAG=${1:-wm1}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
export PRIMESM_HIP_LIB=$PWD/primestereomatch_amd/lib/libprimesm_hip_exp.so
PSM_WM_TRACE=1 timeout 300 python bench.py --pp --steps 3 --warmup 1 --no-cpu-baseline --frame-loop 0 > $O/bench_pp.json 2> $O/bench_pp.err
PSM_WM_TRACE=1 timeout 300 python scripts/dbg_wmf.py big > $O/wmf.txt 2> $O/wmf.err
unset PRIMESM_HIP_LIB
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o pp -- python bench.py --pp --steps 3 --warmup 1 --no-cpu-baseline --frame-loop 0 > $O/prof.log 2>&1
python scripts/wm_trace_summary.py $O > $O/trace_wm.txt 2>&1
grep "\[wm\]" $O/bench_pp.err | tail -2 | cut -c1-1500
python -c "
import json;j=json.loads(open('$O/bench_pp.json').read().strip().splitlines()[-1]);print(j['pp'])"
tail -12 $O/trace_wm.txt
