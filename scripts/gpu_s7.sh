#!/bin/bash
# (libprimesm_hip_prev.so = the library of the commit before, built into lib/ for the same-box A/B: git stash; make OUT=../lib/libprimesm_hip_prev.so OBJDIR=../lib/obj_prev)
# overlap of consecutive slices of a chunk (plane form): correctness + same-box A/B against the library before it
TAG=${1:-r6s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/primestereomatch_amd/lib
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 --deselect tests/test_gpu_bench.py > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
B="timeout 600 python bench.py --no-cpu-baseline --frame-loop 0"
for rep in 1 2; do
for v in new prev; do
  lib=$L/libprimesm_hip.so; [ $v = prev ] && lib=$L/libprimesm_hip_prev.so
  PRIMESM_HIP_LIB=$lib $B --config c2 --pair fixture --steps 50 --warmup 10 > $OUT/c2_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --config c1 --pair fixture --steps 50 --warmup 10 > $OUT/c1_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --config c1x --pair fixture --steps 50 --warmup 10 > $OUT/c1x_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --config c2 --pair fixture --steps 50 --warmup 10 --frames-in-flight 2 > $OUT/c2_fif2_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --config c2 --batch 8 --steps 30 --warmup 5 > $OUT/c2_b8_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --steps 20 --warmup 5 > $OUT/c4_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --config c3 --steps 30 --warmup 5 > $OUT/c3_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --shard-sim 8 --shard disp --steps 40 --no-oracle-check > $OUT/disp8_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --shard-sim 8 --steps 40 --no-oracle-check > $OUT/rows8_${v}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --flags 2097152 --steps 10 > $OUT/c4_single_${v}_$rep.json 2>> $OUT/err.txt
done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in (j["kernels"].get("cvf_fused",{}).get("by_form") or {}).items()}, "verified", j.get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
