#!/bin/bash
# Round-3 session I: seed stride x spread (experiments build)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frame-loop 0"
show() { python - "$@" <<PY
import json,sys
for f in sys.argv[1:]:
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        fl=j["kernels"].get("cvf_fused",{}).get("by_form",{})
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in fl.items()}, {k:v["avg_ms"] for k,v in j["kernels"].items() if k!="cvf_fused"})
    except Exception as e:
        print(f, "ERR", e)
PY
}
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip_exp.so
for S in 5 6 7 8 10 12; do for sp in 4 8; do PSM_PC_S=$S PSM_PC_SPREAD=$sp $B > $OUT/c4_S${S}_sp$sp.json 2>> $OUT/err; done; done
show $OUT/c4_S*.json
for S in 4 5 6 8; do PSM_PC_S=$S $B --config c3 > $OUT/c3_S$S.json 2>> $OUT/err; PSM_PC_S=$S $B --shard-sim 8 --steps 40 > $OUT/s8rows_S$S.json 2>> $OUT/err; PSM_PC_S=$S $B --config c5 --steps 5 --warmup 2 > $OUT/c5_S$S.json 2>> $OUT/err; done
show $OUT/c3_S*.json $OUT/s8rows_S*.json $OUT/c5_S*.json
tail -3 $OUT/err
