#!/bin/bash
# the world-2 same-device lines of scripts/gpu_round.sh alone
TAG=${1:-w2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "
import datetime,subprocess
print('$TAG', datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ'))" > $OUT/device.txt
W2="timeout 900 python bench.py --gpus 2 --same-device --frames-in-flight 2 --steps 10 --warmup 3"
$W2 > $OUT/bench_c4_world2_same_device.json 2> $OUT/bench_world2.err
python - <<PY
import json
j=json.loads(open('$OUT/bench_c4_world2_same_device.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['verified_vs_single_gpu'], j['oracle_maps_equal'], j['config']['frames_in_flight'], j['alt_shard']['ms_per_step'], j['alt_shard']['verified_vs_single_gpu'])
PY
