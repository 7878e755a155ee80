#!/bin/bash
# short session: smoke, the tests this round touched, the default line, the distributed path with two frames in flight
TAG=${1:-r6a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ): $(rocminfo 2>/dev/null | grep -m1 -oE 'gfx9[0-9a-f]+'), host $(grep -m1 'model name' /proc/cpuinfo | sed 's/.*: //'), $(nproc) cores, box $(hostname)" > $OUT/device.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_fma.py tests/test_gpu_api.py tests/test_gpu_stripes.py tests/test_gpu_bench.py tests/test_capi_host.py -m gpu -q -x -p no:cacheprovider --timeout=900 > $OUT/pytest_sel.log 2>&1
tail -5 $OUT/pytest_sel.log
B="timeout 600 python bench.py"
$B --no-cpu-wide > $OUT/bench_c4_n1.json 2> $OUT/bench.err
$B --shard-sim 8 --steps 40 --no-oracle-check > $OUT/bench_c4_shardsim_1of8.json 2>> $OUT/bench.err
$B --shard-sim 8 --steps 40 --frames-in-flight 2 --no-oracle-check > $OUT/bench_c4_shardsim_1of8_fif2.json 2>> $OUT/bench.err
$B --shard-sim 8 --shard disp --steps 40 --no-oracle-check > $OUT/bench_c4_shardsim_disp_1of8.json 2>> $OUT/bench.err
$B --config c2 --pair fixture --steps 30 --no-cpu-wide > $OUT/bench_c2_teddy_n1.json 2>> $OUT/bench.err
D="timeout 600 python bench.py --gpus 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline"
$D > $OUT/bench_c4_dist_world1.json 2> $OUT/bench_dist1.err
$D --frames-in-flight 1 > $OUT/bench_c4_dist_world1_fif1.json 2>> $OUT/bench_dist1.err
W2="timeout 900 python bench.py --gpus 2 --same-device --steps 10 --warmup 3"
$W2 > $OUT/bench_c4_world2_same_device.json 2> $OUT/bench_world2.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.3e vox/s"%j["value"], "%.4f ms"%j["ms_per_step"], {k:round(v["avg_ms"],4) for k,v in j["kernels"].items()}, "fif", j["config"].get("frames_in_flight"), "verified", j.get("verified_vs_single_gpu"), "oracle", j.get("oracle_maps_equal"), "fifeq", j.get("frames_in_flight_maps_equal"), "alt", (j.get("alt_shard") or {}).get("ms_per_step"), (j.get("alt_shard") or {}).get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/bench.err $OUT/bench_dist1.err $OUT/bench_world2.err
echo "== done"
