#!/bin/bash
TAG=${1:-r02e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for dc in 1 2 4 8 16; do for seg in 0 135 90 68; do
  PSM_PC_DC=$dc timeout 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --seg-rows $seg > $OUT/b_${dc}_${seg}.json 2>> $OUT/err.log
done; done
for seg in 0 270 135 90; do timeout 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --seg-rows $seg --flags 8192 > $OUT/s_${seg}.json 2>> $OUT/err.log; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "%.3f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items() if k in ("cvf_fused","wta")})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/err.log
