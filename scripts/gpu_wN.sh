#!/bin/bash
# the N > 1 protocol with 4 and 8 ranks on ONE GPU (gloo-staged exchange): what `bench.py --gpus 4 / 8` runs by default
# (two frames in flight per rank from N = 4), verified against the single-GPU maps and the oracle - correctness lines, not timings
TAG=${1:-wN}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for n in 4 8; do
  timeout 1200 python bench.py --gpus $n --same-device --steps 6 --warmup 2 > $OUT/bench_c4_world${n}_same_device.json 2> $OUT/bench_world$n.err
  timeout 1200 python bench.py --gpus $n --same-device --shard disp --steps 6 --warmup 2 > $OUT/bench_c4_world${n}_same_device_disp.json 2>> $OUT/bench_world$n.err
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        a=j['alt_shard']
        print(os.path.basename(f), 'ranks', j['ranks'], 'fif', j['config']['frames_in_flight'], a['frames_in_flight'], 'verified', j['verified_vs_single_gpu'], a['verified_vs_single_gpu'], 'oracle', j.get('oracle_maps_equal'), a.get('oracle_maps_equal'), 'fifeq', j.get('frames_in_flight_maps_equal'), round(j['ms_per_step'],2), j['exchange_backend'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
grep -h "bench.py:" $OUT/*.err | sort | uniq -c | head
