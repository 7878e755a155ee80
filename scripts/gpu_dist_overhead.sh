#!/bin/bash
# fixed per-step cost of the torch.distributed path: world-1 RCCL runs of small configurations against the plain path
TAG=${1:-distov}
O=gpurun_out/$TAG
mkdir -p $O
for cfg in c3 c2; do
  python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --frame-loop 0 > $O/${cfg}_plain.json 2>> $O/err.txt
  for f in 1 2; do
    python bench.py --config $cfg --gpus 1 --force-dist --frames-in-flight $f --steps 60 --warmup 10 --no-cpu-baseline > $O/${cfg}_dist_fif$f.json 2>> $O/err.txt
    python bench.py --config $cfg --gpus 1 --force-dist --shard disp --frames-in-flight $f --steps 60 --warmup 10 --no-cpu-baseline > $O/${cfg}_dist_disp_fif$f.json 2>> $O/err.txt
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(j['ms_per_step'],4), j.get('per_rank',{}).get('compute_ms'), j.get('per_rank',{}).get('collective_ms'))
    except Exception as e: print(os.path.basename(f),'ERR',e)
PY
tail -3 $O/err.txt
