#!/bin/bash
# Round-3 session W: weighted median - parity (all post-processing tests, fuzz), timing of the synthetic cases and of the bench pair
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider --timeout=1200 -k "wmf or wgt or median or pp or process_dm or fuzz" 2>&1 | grep -E "long_lists|230x110|passed|failed" | tail -8
timeout 300 python scripts/dbg_wmf.py big 2>&1 | tee $OUT/wmf.txt | tail -12
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --frame-loop 0 --pp > $OUT/c4pp.json 2>$OUT/err; python -c "
import json;j=json.loads([l for l in open('$OUT/c4pp.json') if l.startswith('{')][-1]);print(j['pp'])"
timeout 200 python scripts/soak.py 60 63 | tail -2
