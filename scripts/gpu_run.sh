#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+variants), rocprofv3 kernel trace.
# Usage (from the repo root, via gpurun): bash scripts/gpu_run.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout=600 > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
echo "== bench c4"
timeout 600 python bench.py --steps 10 --warmup 3 --box-bench > $OUT/bench_c4.json 2> $OUT/bench_c4.err; cat $OUT/bench_c4.json; tail -3 $OUT/bench_c4.err
for w in 1 2 8; do
  timeout 300 python bench.py --steps 5 --warmup 2 --waves $w --no-cpu-baseline > $OUT/bench_c4_w$w.json 2>> $OUT/bench_var.err
done
for sr in 135 270 540; do
  timeout 300 python bench.py --steps 5 --warmup 2 --seg-rows $sr --no-cpu-baseline > $OUT/bench_c4_sr$sr.json 2>> $OUT/bench_var.err
done
timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c3.json 2>> $OUT/bench_var.err
timeout 300 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench_var.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "%.3e vox/s"%j["value"], "%.2f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, j["roofline"]["frac"], j.get("box_filter_pass"))
    except Exception as e:
        print(f, "ERR", e)
PY
echo "== torch.distributed path on 1 GPU (RCCL all-gather with world_size 1)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
cat $OUT/bench_dist1.json; tail -5 $OUT/bench_dist1.err
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
# keep the merged output small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"
