#!/bin/bash
# Round-3 session H: suite after spread order / planner / 8-bit trims, then the affected bench lines
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frame-loop 0"
show() { python - "$@" <<PY
import json,sys
for f in sys.argv[1:]:
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        fl=j["kernels"].get("cvf_fused",{}).get("by_form",{})
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in fl.items()}, {k:v["avg_ms"] for k,v in j["kernels"].items() if k!="cvf_fused"}, j.get("verified_vs_single_gpu"), "frac", j["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
}
$B --verify > $OUT/c4_f32.json 2>> $OUT/err; $B --dtype u8 --verify > $OUT/c4_u8.json 2>> $OUT/err
for c in c3 c2 c1 c1x; do $B --config $c --steps 50 --verify > $OUT/${c}.json 2>> $OUT/err; done
for g in 2 4 8; do $B --shard-sim $g --steps 40 > $OUT/s${g}rows.json 2>> $OUT/err; $B --shard-sim $g --shard disp --steps 40 > $OUT/s${g}disp.json 2>> $OUT/err; done
show $OUT/c4_f32.json $OUT/c4_u8.json $OUT/c3.json $OUT/c2.json $OUT/c1.json $OUT/c1x.json $OUT/s*rows.json $OUT/s*disp.json
tail -3 $OUT/err
