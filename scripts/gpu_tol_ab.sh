#!/bin/bash
# the tolerance-form A/B lines of scripts/gpu_round.sh alone (same commands), into gpurun_out/<tag>
TAG=${1:-tolab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python - <<PY > $OUT/device.txt
import subprocess,datetime
print("$TAG", datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"))
PY
B="python bench.py"
for rep in 1 2; do
  $B --steps 20 --warmup 5 --no-cpu-baseline --frame-loop 0 > $OUT/bench_c4_exact_ab$rep.json 2>> $OUT/err.txt
  $B --steps 20 --warmup 5 --flags 33554432 --no-cpu-wide --frame-loop 0 > $OUT/bench_c4_tol_ab$rep.json 2>> $OUT/err.txt
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j['roofline']; print(os.path.basename(f), round(j['ms_per_step'],4), r['frac'], r.get('exceeds_hbm_peak'), r['frac_basis'][:30])
PY
