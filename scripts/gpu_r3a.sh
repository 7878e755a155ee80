#!/bin/bash
# Round-3 session A: parity tests on the new k_cvf_pc (two batches per loop iteration, scaled sums), then same-box A/B
# of the library variants.  Usage (via gpurun): bash scripts/gpu_r3a.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=900 > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
L=$GRAFT_REPO_ROOT/primestereomatch_amd/lib
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5"
for rep in 1 2; do
for v in r02 u2only sconly new; do
  lib=$L/libprimesm_hip_$v.so; [ $v = new ] && lib=$L/libprimesm_hip.so
  PRIMESM_HIP_LIB=$lib $B --verify > $OUT/ab_c4_${v}_$rep.json 2>> $OUT/ab.err
done; done
for v in r02 new; do
  lib=$L/libprimesm_hip_$v.so; [ $v = new ] && lib=$L/libprimesm_hip.so
  PRIMESM_HIP_LIB=$lib $B --config c3 > $OUT/ab_c3_$v.json 2>> $OUT/ab.err
  PRIMESM_HIP_LIB=$lib $B --config c2 --steps 50 > $OUT/ab_c2_$v.json 2>> $OUT/ab.err
  PRIMESM_HIP_LIB=$lib $B --config c1 --steps 50 > $OUT/ab_c1_$v.json 2>> $OUT/ab.err
  PRIMESM_HIP_LIB=$lib $B --config c4 --dtype u8 > $OUT/ab_c4u8_$v.json 2>> $OUT/ab.err
  PRIMESM_HIP_LIB=$lib $B --shard-sim 8 --steps 40 > $OUT/ab_s8rows_$v.json 2>> $OUT/ab.err
  PRIMESM_HIP_LIB=$lib $B --shard-sim 8 --shard disp --steps 40 > $OUT/ab_s8disp_$v.json 2>> $OUT/ab.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.3f ms"%j["ms_per_step"], "median %.3f"%j["median_ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "verified", j.get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/ab.err
