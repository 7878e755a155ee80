#!/bin/bash
# Round-3 session L: channel-split guidance kernel - parity + times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 2>&1 | tail -4
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frame-loop 0"
for c in c4 c3 c2 c1x c5; do $B --config $c > $OUT/$c.json 2>> $OUT/err; done
$B --shard-sim 8 --steps 40 > $OUT/s8rows.json 2>> $OUT/err; $B --shard-sim 8 --shard disp --steps 40 > $OUT/s8disp.json 2>> $OUT/err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
    print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "frac", j["roofline"]["frac"])
PY
tail -2 $OUT/err
