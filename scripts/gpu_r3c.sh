#!/bin/bash
# Round-3 session C: GPU suite after the bench rewrite / host mirror fixes + the new bench line.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=1200 > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
timeout 600 python bench.py --verify --box-bench --pp > $OUT/bench_c4.json 2> $OUT/bench.err; cat $OUT/bench_c4.json; tail -5 $OUT/bench.err
timeout 600 python bench.py --gpus 1 --force-dist --no-cpu-baseline > $OUT/bench_c4_dist1.json 2>> $OUT/bench.err; cat $OUT/bench_c4_dist1.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['verified_vs_single_gpu'], j['alt_shard'])"
