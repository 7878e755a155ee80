#!/bin/bash
# what binds the lane-per-pixel evaluation?  diagnostic builds without the weight loads / the map loads / the LDS histogram
TAG=${1:-wmdbg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in "" _dbg_COAL; do
  export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip$v.so
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof$v -o t -- python $GRAFT_REPO_ROOT/scripts/dbg_wmf.py hd20 > $OUT/run$v.log 2>&1
  fdb=$(find $OUT/prof$v -name "*.db" | head -1)
  echo "== variant '$v'"; tail -1 $OUT/run$v.log | cut -c1-200
  [ -n "$fdb" ] && python - <<PY
import sqlite3
rows=list(sqlite3.connect("$fdb").execute("select name,start,end from kernels order by start"))
ev=[(e-s)/1e3 for n,s,e in rows if 'k_wm_eval<' in n]
print('k_wm_eval launches (us):', [round(x,1) for x in ev[:14]])
PY
done
