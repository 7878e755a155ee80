#!/bin/bash
# bench the prebuilt library variants primestereomatch_amd/lib/libprimesm_hip_*.so
# bash scripts/gpu_variants.sh <tag> "<bench args>"
TAG=${1:-v}; ARGS=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in primestereomatch_amd/lib/libprimesm_hip*.so; do
  n=$(basename $lib .so)
  PRIMESM_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --box-bench $ARGS > $OUT/$n.json 2> $OUT/$n.err || tail -3 $OUT/$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/$n.json").read().strip().splitlines()[-1])
    print("[$n $ARGS]", "%.3e vox/s"%j["value"], "%.2f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "box", (j.get("box_filter_pass") or {}).get("avg_ms"))
except Exception as e:
    print("[$n] ERR", e)
PY
done
