#!/bin/bash
# per-rank time of a 1/G share with one and two frames in flight (which should `bench.py --gpus N` run?)
TAG=${1:-fif}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --no-cpu-baseline --frame-loop 0 --steps 40 --warmup 5"
for rep in 1 2; do
for g in 2 4 8; do for f in 1 2; do
  $B --shard-sim $g --frames-in-flight $f > $O/rows_1of${g}_fif${f}_$rep.json 2>> $O/err.txt
done; done; done
for g in 2 4; do for f in 1 2; do
  $B --shard-sim $g --shard disp --frames-in-flight $f > $O/disp_1of${g}_fif${f}_1.json 2>> $O/err.txt
done; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json')):
    try: j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j['ms_per_step'],4))
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
