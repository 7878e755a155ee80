#!/bin/bash
# short session: the two --shard-sim 8 --frames-in-flight 2 lines of scripts/gpu_round.sh + tests/test_gpu_bench.py (after the None comparison fix in bench.py)
TAG=${1:-fix}
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="timeout 600 python bench.py"
$B --shard-sim 8 --frames-in-flight 2 --steps 40 > $OUT/bench_c4_shardsim_1of8_fif2.json 2>> $OUT/err.txt
$B --shard-sim 8 --shard disp --frames-in-flight 2 --steps 40 > $OUT/bench_c4_shardsim_disp_1of8_fif2.json 2>> $OUT/err.txt
timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -3
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j['ms_per_step'],4), j['roofline'].get('frac'), j.get('verified_vs_single_gpu'), j.get('frames_in_flight_maps_equal'))
PY
