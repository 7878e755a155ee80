#!/bin/bash
TAG=${1:-r02c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
for cfg in "" "--flags 524288" "--flags 8192"; do
  n=$(echo "$cfg" | tr -d ' -')
  timeout 300 python bench.py --no-cpu-baseline --verify $cfg > $OUT/bench_c4_$n.json 2>> $OUT/bench.err
done
for g in 2 4 8; do timeout 300 python bench.py --shard-sim $g --no-cpu-baseline > $OUT/bench_c4_shardsim_$g.json 2>> $OUT/bench.err; done
timeout 300 python bench.py --config c3 --no-cpu-baseline --verify > $OUT/bench_c3.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c2 --steps 30 --no-cpu-baseline --verify > $OUT/bench_c2.json 2>> $OUT/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "%.3e vox/s"%j["value"], "%.3f ms"%j["ms_per_step"], "med %.3f"%j["median_ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "frac", j["roofline"]["frac"], "verified", j.get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $OUT/bench.err
for cfg in "--config c1" "--config c1x" "--config c4 --dtype u8"; do
  n=$(echo "$cfg" | tr -d ' -')
  timeout 300 python bench.py --no-cpu-baseline --verify --steps 20 $cfg > $OUT/bench_u8_$n.json 2>> $OUT/bench.err
  python -c "
import json,sys
j=json.loads(open('$OUT/bench_u8_$n.json').read().strip().splitlines()[-1])
print('u8', '$n', '%.3e vox/s'%j['value'], '%.3f ms'%j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()}, 'frac', j['roofline']['frac'], 'verified', j.get('verified_vs_single_gpu'))"
done
tail -3 $OUT/bench.err
