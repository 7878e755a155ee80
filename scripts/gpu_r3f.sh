#!/bin/bash
# Round-3 session F: spread order of the key phase (f32 / u8), segment / DC sweeps at c4 and c3 per phase
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
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
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in fl.items()}, {k:v["avg_ms"] for k,v in j["kernels"].items() if k!="cvf_fused"}, j.get("verified_vs_single_gpu"))
    except Exception as e:
        print(f, "ERR", e)
PY
}
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip_exp.so
echo "== spread"
for sp in 1 2 4 8 16; do PSM_PC_SPREAD=$sp $B --verify > $OUT/c4_f32_sp$sp.json 2>> $OUT/err; PSM_PC_SPREAD=$sp $B --dtype u8 --verify > $OUT/c4_u8_sp$sp.json 2>> $OUT/err; done
show $OUT/c4_f32_sp*.json $OUT/c4_u8_sp*.json
echo "== c4 segments x DC"
for dc in 1 2; do for sr in 0 120 135 180 216 270 360 540; do PSM_PC_DC=$dc $B --seg-rows $sr > $OUT/c4_sr${sr}_dc$dc.json 2>> $OUT/err; done; done
show $OUT/c4_sr*.json
echo "== c3 segments x DC"
for dc in 1 2; do for sr in 0 90 120 144 180 240 360; do PSM_PC_DC=$dc $B --config c3 --seg-rows $sr > $OUT/c3_sr${sr}_dc$dc.json 2>> $OUT/err; done; done
show $OUT/c3_sr*.json
echo "== 1/8 stripe segments x DC"
for dc in 1 2; do for sr in 0 45 68 135; do PSM_PC_DC=$dc $B --shard-sim 8 --steps 40 --seg-rows $sr > $OUT/s8rows_sr${sr}_dc$dc.json 2>> $OUT/err; done; done
show $OUT/s8rows_sr*.json
tail -3 $OUT/err
