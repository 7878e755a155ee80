#!/bin/bash
# Round-3 session E: where the 8-bit mode and the small configs lose time (per-form launch times), stride / segment sweeps
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
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
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in fl.items()}, "frac", j["roofline"]["frac"], {k:v["avg_ms"] for k,v in j["kernels"].items() if k!="cvf_fused"})
    except Exception as e:
        print(f, "ERR", e)
PY
}
$B > $OUT/c4_f32.json 2>> $OUT/err; $B --dtype u8 > $OUT/c4_u8.json 2>> $OUT/err
for c in c3 c2 c1 c1x; do $B --config $c --steps 50 > $OUT/${c}.json 2>> $OUT/err; done
show $OUT/c4_f32.json $OUT/c4_u8.json $OUT/c3.json $OUT/c2.json $OUT/c1.json $OUT/c1x.json
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip_exp.so
echo "== stride sweep f32 / u8 (experiments build)"
for S in 3 4 5 6 8 10; do PSM_PC_S=$S $B > $OUT/c4_f32_S$S.json 2>> $OUT/err; PSM_PC_S=$S $B --dtype u8 > $OUT/c4_u8_S$S.json 2>> $OUT/err; done
show $OUT/c4_f32_S*.json $OUT/c4_u8_S*.json
echo "== c2 / c1 segment + DC sweep"
for sr in 0 47 63 94 125 188 375; do for dc in 1 2 4; do PSM_PC_DC=$dc $B --config c2 --steps 50 --seg-rows $sr > $OUT/c2_sr${sr}_dc$dc.json 2>> $OUT/err; done; done
show $OUT/c2_sr*.json
for dc in 1 2 4; do for sr in 0 72 96 144 288; do PSM_PC_DC=$dc $B --config c1x --steps 50 --seg-rows $sr > $OUT/c1x_sr${sr}_dc$dc.json 2>> $OUT/err; done; done
show $OUT/c1x_sr*.json
echo "== D-shard of 32 slices: DC / segments"
for dc in 1 2; do for sr in 0 135 180 270 360; do PSM_PC_DC=$dc $B --shard-sim 8 --shard disp --steps 40 --seg-rows $sr > $OUT/s8disp_sr${sr}_dc$dc.json 2>> $OUT/err; done; done
show $OUT/s8disp_sr*.json
for f in 1048576; do $B --shard-sim 8 --shard disp --steps 40 --flags $f > $OUT/s8disp_twophase.json 2>> $OUT/err; done; show $OUT/s8disp_twophase.json
tail -3 $OUT/err
