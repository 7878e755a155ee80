#!/bin/bash
# after the planner change (S = 8, key-form cost model): all GPU tests + the lines of every config; two-phase threshold check at 64 slices
TAG=${1:-r6i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ) box $(hostname)" > $OUT/device.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 -x > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
B="timeout 600 python bench.py --no-cpu-wide --frame-loop 0"
$B > $OUT/c4.json 2>> $OUT/err.txt
$B --config c3 --steps 30 > $OUT/c3.json 2>> $OUT/err.txt
$B --config c5 --steps 4 --warmup 1 > $OUT/c5.json 2>> $OUT/err.txt
$B --dtype u8 > $OUT/c4_u8.json 2>> $OUT/err.txt
$B --config c2 --pair fixture --steps 50 > $OUT/c2.json 2>> $OUT/err.txt
$B --config c2 --pair fixture --steps 50 --flags 1048576 > $OUT/c2_two.json 2>> $OUT/err.txt
$B --config c1 --pair fixture --steps 50 > $OUT/c1.json 2>> $OUT/err.txt
$B --config c1 --pair fixture --steps 50 --flags 1048576 > $OUT/c1_two.json 2>> $OUT/err.txt
$B --config c3 --steps 30 --frames-in-flight 2 > $OUT/c3_fif2.json 2>> $OUT/err.txt
$B --shard-sim 8 --steps 40 > $OUT/rows8.json 2>> $OUT/err.txt
$B --shard-sim 8 --steps 40 --frames-in-flight 2 > $OUT/rows8_fif2.json 2>> $OUT/err.txt
$B --shard-sim 4 --steps 30 > $OUT/rows4.json 2>> $OUT/err.txt
$B --shard-sim 2 --steps 20 > $OUT/rows2.json 2>> $OUT/err.txt
$B --shard-sim 2 --shard disp --steps 20 > $OUT/disp2.json 2>> $OUT/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:round(v["avg_ms"],4) for k,v in j["kernels"].items() if k!="cvf_fused"}, {k:v["avg_ms"] for k,v in (j["kernels"].get("cvf_fused",{}).get("by_form") or {}).items()}, "verified", j.get("verified_vs_single_gpu"), "oracle", j.get("oracle_maps_equal"), "pf", j["roofline"].get("pipeline_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $OUT/err.txt
