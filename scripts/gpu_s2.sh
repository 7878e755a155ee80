#!/bin/bash
# session: the in-kernel reduction - correctness (new tests + the parity suites through the default path) and same-box A/B
TAG=${1:-r6b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ): $(rocminfo 2>/dev/null | grep -m1 -oE 'gfx9[0-9a-f]+'), host $(grep -m1 'model name' /proc/cpuinfo | sed 's/.*: //'), $(nproc) cores, box $(hostname)" > $OUT/device.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== tests (fuse first)"
timeout 900 python -m pytest tests/test_gpu_fuse.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $OUT/pytest_fuse.log 2>&1; tail -4 $OUT/pytest_fuse.log
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=1200 --deselect tests/test_gpu_fuse.py > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
B="timeout 600 python bench.py --no-cpu-wide"
for rep in 1 2; do
  for f in 1 0; do
    $B --fused-reduce $f --steps 20 --warmup 5 --frame-loop 0 > $OUT/bench_c4_fuse${f}_ab$rep.json 2>> $OUT/bench.err
    $B --fused-reduce $f --config c3 --steps 30 --warmup 5 --frame-loop 0 > $OUT/bench_c3_fuse${f}_ab$rep.json 2>> $OUT/bench.err
    $B --fused-reduce $f --config c2 --pair fixture --steps 50 --warmup 10 --frame-loop 0 > $OUT/bench_c2_teddy_fuse${f}_ab$rep.json 2>> $OUT/bench.err
    $B --fused-reduce $f --config c1 --pair fixture --steps 50 --warmup 10 --frame-loop 0 > $OUT/bench_c1_cones_fuse${f}_ab$rep.json 2>> $OUT/bench.err
    $B --fused-reduce $f --shard-sim 8 --steps 40 --no-oracle-check > $OUT/bench_c4_shardsim_1of8_fuse${f}_ab$rep.json 2>> $OUT/bench.err
    $B --fused-reduce $f --shard-sim 8 --shard disp --steps 40 --no-oracle-check > $OUT/bench_c4_shardsim_disp_1of8_fuse${f}_ab$rep.json 2>> $OUT/bench.err
  done
done
$B --fused-reduce 2 --steps 20 --warmup 5 --frame-loop 0 > $OUT/bench_c4_fuse2.json 2>> $OUT/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], "med %.4f"%j["median_ms_per_step"], {k:round(v["avg_ms"],4) for k,v in j["kernels"].items()}, {k:v["avg_ms"] for k,v in (j["kernels"].get("cvf_fused",{}).get("by_form") or {}).items()}, "verified", j.get("verified_vs_single_gpu"), "oracle", j.get("oracle_maps_equal"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/bench.err
echo "== rocprofv3 kernel trace of the default line and of c2"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --frame-loop 0 > $OUT/rocprof_stdout.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f 20 > $OUT/trace_gaps_c4.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -o trace -- python $GRAFT_REPO_ROOT/bench.py --config c2 --pair fixture --steps 30 --warmup 3 --no-cpu-baseline --frame-loop 0 > $OUT/rocprof_c2_stdout.log 2>&1
f=$(find $OUT/prof_c2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f 20 > $OUT/trace_gaps_c2.txt 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
cat $OUT/trace_gaps_c4.txt | head -24; cat $OUT/trace_gaps_c2.txt | head -24
echo "== done"
