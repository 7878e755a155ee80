#!/bin/bash
# Round-3 session K: why is the key phase slow on a 32-slice disparity shard?
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --frame-loop 0 --shard-sim 8 --shard disp"
show() { python - "$@" <<PY
import json,sys
for f in sys.argv[1:]:
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        fl=j["kernels"].get("cvf_fused",{}).get("by_form",{})
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:(v["avg_ms"],v["min_ms"]) for k,v in fl.items()}, {k:v["avg_ms"] for k,v in j["kernels"].items() if k!="cvf_fused"})
    except Exception as e:
        print(f, "ERR", e)
PY
}
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip_exp.so
$B > $OUT/base.json 2>> $OUT/err
for S in 2 3 4 5 8; do for sr in 0 270 540 1080; do PSM_PC_S=$S $B --flags 1048576 --seg-rows $sr > $OUT/tp_S${S}_sr$sr.json 2>> $OUT/err; done; done
show $OUT/base.json $OUT/tp_*.json
for sp in 1 2 4; do PSM_PC_SPREAD=$sp PSM_PC_S=4 $B --flags 1048576 > $OUT/tp_S4_sp$sp.json 2>> $OUT/err; done
show $OUT/tp_S4_sp*.json
tail -3 $OUT/err
