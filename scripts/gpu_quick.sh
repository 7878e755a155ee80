#!/bin/bash
# Fast iteration loop on the GPU box: build, parity tests, a handful of bench variants.
# bash scripts/gpu_quick.sh <tag> ["bench args variant 1" "bench args variant 2" ...]
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo "BUILD FAILED"; tail -20 $OUT/build.log; }
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
i=0
for v in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --box-bench $v > $OUT/bench_$i.json 2> $OUT/bench_$i.err || tail -3 $OUT/bench_$i.err
  python - <<PY
import json
try:
    j=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
    print("[$v]", "%.3e vox/s"%j["value"], "%.2f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "frac", j["roofline"]["frac"], "box", (j.get("box_filter_pass") or {}).get("avg_ms"))
except Exception as e:
    print("[$v] ERR", e)
PY
done
