#!/bin/bash
# Round-3 session V: maps written by the single-phase reduction - full GPU suite, state-machine soak, small-config A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 2>&1 | tail -4
timeout 200 python scripts/soak_state.py 90 7 2>&1 | tail -3
B="timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --frame-loop 0"
for rep in 1 2 3; do
for lib in new prev; do
  if [ $lib = prev ]; then export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip_prev.so; else unset PRIMESM_HIP_LIB; fi
  for c in c2 c1 c1x; do $B --config $c > $OUT/$lib.$c.$rep.json 2>> $OUT/err; done
done; done
python - <<PY
import json,glob,collections
d=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        lib,c,rep=f.split('/')[-1].split('.')[:3]
        d[(c,lib)].append(j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
for k in sorted(d): print(k, ["%.4f"%v for v in d[k]], "min %.4f"%min(d[k]))
PY
tail -3 $OUT/err
