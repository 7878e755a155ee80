#!/bin/bash
TAG=${1:-r02h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scripts/dbg_q2.py 2>&1 | grep -v "mismatch 0" | head
for cfg in "" "--flags 16384" "--seg-rows 270" "--seg-rows 135"; do
  n=$(echo "$cfg" | tr -d ' -')
  timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 $cfg > $OUT/bench_c4_$n.json 2>> $OUT/bench.err
done
for dc in 1 4; do PSM_PC_DC=$dc timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_c4_dc$dc.json 2>> $OUT/bench.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "%.3f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items() if k in ("cvf_fused","wta")})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/bench.err
