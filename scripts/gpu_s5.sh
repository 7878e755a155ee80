#!/bin/bash
# the plane form's in-kernel reduction, second version (no LDS parking, nothing in the key form, eight chunks in flight per thread)
TAG=${1:-r6e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/primestereomatch_amd/lib
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ) box $(hostname)" > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_fuse.py -m gpu -q -x -p no:cacheprovider --timeout=600 > $OUT/pytest_fuse.log 2>&1; tail -4 $OUT/pytest_fuse.log
B="timeout 600 python bench.py --no-cpu-baseline --frame-loop 0"
for rep in 1 2; do
  PRIMESM_HIP_LIB=$L/libprimesm_hip_notail.so $B --fused-reduce 0 --steps 20 --warmup 5 > $OUT/c4_notail_$rep.json 2>> $OUT/err.txt
  for f in 0 1; do
    $B --fused-reduce $f --steps 20 --warmup 5 > $OUT/c4_fuse${f}_$rep.json 2>> $OUT/err.txt
    $B --fused-reduce $f --config c3 --steps 30 --warmup 5 > $OUT/c3_fuse${f}_$rep.json 2>> $OUT/err.txt
    $B --fused-reduce $f --config c2 --pair fixture --steps 50 --warmup 10 > $OUT/c2_fuse${f}_$rep.json 2>> $OUT/err.txt
    $B --fused-reduce $f --config c1 --pair fixture --steps 50 --warmup 10 > $OUT/c1_fuse${f}_$rep.json 2>> $OUT/err.txt
    $B --fused-reduce $f --shard-sim 8 --steps 40 --no-oracle-check > $OUT/rows8_fuse${f}_$rep.json 2>> $OUT/err.txt
    $B --fused-reduce $f --shard-sim 8 --shard disp --steps 40 --no-oracle-check > $OUT/disp8_fuse${f}_$rep.json 2>> $OUT/err.txt
    $B --fused-reduce $f --config c5 --steps 4 --warmup 1 --no-oracle-check > $OUT/c5_fuse${f}_$rep.json 2>> $OUT/err.txt
  done
  PRIMESM_HIP_LIB=$L/libprimesm_hip_notail.so $B --fused-reduce 0 --config c2 --pair fixture --steps 50 --warmup 10 > $OUT/c2_notail_$rep.json 2>> $OUT/err.txt
done
$B --fused-reduce 2 --steps 20 --warmup 5 > $OUT/c4_fuse2.json 2>> $OUT/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:round(v["avg_ms"],4) for k,v in j["kernels"].items() if k!="cvf_fused"}, {k:v["avg_ms"] for k,v in (j["kernels"].get("cvf_fused",{}).get("by_form") or {}).items()}, "verified", j.get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $OUT/err.txt
