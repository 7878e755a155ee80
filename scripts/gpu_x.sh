#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 --warmup 3 --verify 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['verified_vs_single_gpu'])"
