#!/bin/bash
# which part of the in-kernel reduction slows the key phase down? library variants, same box
TAG=${1:-r6c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/primestereomatch_amd/lib
B="timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frame-loop 0"
for rep in 1 2; do
for v in v0 v1 v3 main; do
  lib=$L/libprimesm_hip_$v.so; [ $v = main ] && lib=$L/libprimesm_hip.so
  PRIMESM_HIP_LIB=$lib $B --fused-reduce 0 > $OUT/c4_${v}_fuse0_$rep.json 2>> $OUT/err.txt
  if [ $v = v1 ] || [ $v = main ]; then PRIMESM_HIP_LIB=$lib $B --fused-reduce 1 > $OUT/c4_${v}_fuse1_$rep.json 2>> $OUT/err.txt; fi
done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/c4_*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in (j["kernels"].get("cvf_fused",{}).get("by_form") or {}).items()}, "verified", j.get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/err.txt
