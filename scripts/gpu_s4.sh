#!/bin/bash
# (round 6 experiment session: strided shards; the libprimesm_hip_kaux{0,1}.so variants were built with -DPSM_KEY_AUX=0/1 - the
# macro is gone now that the policy is settled, profiles/r06/exp_key_load_policy.txt)
# experiments: strided disparity shards (two-phase on a 32-slice shard), key-load cache policy at 4K / 1080p; tests of the strided path
TAG=${1:-r6d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/primestereomatch_amd/lib
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ) box $(hostname)" > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_fuse.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "strided" > $OUT/pytest_strided.log 2>&1; tail -4 $OUT/pytest_strided.log
B="timeout 600 python bench.py --no-cpu-baseline --frame-loop 0"
$B --steps 20 --warmup 5 > $OUT/c4_ref.json 2>> $OUT/err.txt
for rep in 1 2; do
  $B --shard-sim 8 --shard disp --steps 40 > $OUT/disp8_contig_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 8 --shard disp --strided --steps 40 > $OUT/disp8_strided_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 8 --shard disp --strided --flags 1048576 --steps 40 > $OUT/disp8_strided_two_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 8 --shard disp --flags 1048576 --steps 40 > $OUT/disp8_contig_two_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 8 --shard disp --strided --flags 1048576 --frames-in-flight 2 --steps 40 > $OUT/disp8_strided_two_fif2_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 4 --shard disp --strided --flags 1048576 --steps 30 > $OUT/disp4_strided_two_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 4 --shard disp --steps 30 > $OUT/disp4_contig_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 2 --shard disp --strided --steps 20 > $OUT/disp2_strided_$rep.json 2>> $OUT/err.txt
  $B --shard-sim 2 --shard disp --steps 20 > $OUT/disp2_contig_$rep.json 2>> $OUT/err.txt
done
for rep in 1 2; do
for a in 16 0 1; do
  lib=$L/libprimesm_hip_kaux$a.so; [ $a = 16 ] && lib=$L/libprimesm_hip.so
  PRIMESM_HIP_LIB=$lib $B --config c5 --steps 4 --warmup 1 --no-oracle-check > $OUT/c5_kaux${a}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --steps 20 --warmup 5 > $OUT/c4_kaux${a}_$rep.json 2>> $OUT/err.txt
  PRIMESM_HIP_LIB=$lib $B --config c3 --steps 30 --warmup 5 > $OUT/c3_kaux${a}_$rep.json 2>> $OUT/err.txt
done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.4f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in (j["kernels"].get("cvf_fused",{}).get("by_form") or {}).items()}, "verified", j.get("verified_vs_single_gpu"), "oracle", j.get("oracle_maps_equal"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $OUT/err.txt
