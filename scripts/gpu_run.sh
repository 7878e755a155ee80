#!/bin/bash
# One full GPU-box session: parity tests, smoke, bench (+ other configs), torch.distributed path on one GPU,
# rocprofv3 kernel trace + HBM-traffic counter passes.  Usage (via gpurun): bash scripts/gpu_run.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || echo "BUILD FAILED"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
echo "== bench (default: c4, N=1)"
timeout 900 python bench.py --box-bench > $OUT/bench_c4.json 2> $OUT/bench_c4.err; cat $OUT/bench_c4.json; tail -2 $OUT/bench_c4.err
timeout 300 python bench.py --config c3 --no-cpu-baseline --box-bench > $OUT/bench_c3.json 2>> $OUT/bench_var.err
timeout 300 python bench.py --config c2 --steps 30 --no-cpu-baseline --box-bench > $OUT/bench_c2.json 2>> $OUT/bench_var.err
timeout 600 python bench.py --config c5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2>> $OUT/bench_var.err
timeout 300 python bench.py --flags 16 --no-cpu-baseline > $OUT/bench_c4_twostage.json 2>> $OUT/bench_var.err
timeout 300 python bench.py --flags 512 --no-cpu-baseline > $OUT/bench_c4_pc2.json 2>> $OUT/bench_var.err
for s in 2 4 8; do timeout 300 python bench.py --fgf $s --no-cpu-baseline > $OUT/bench_c4_fgf_s$s.json 2>> $OUT/bench_var.err; done
for g in 2 4 8; do timeout 300 python bench.py --shard-sim $g --no-cpu-baseline > $OUT/bench_c4_shardsim_$g.json 2>> $OUT/bench_var.err; done
echo "== torch.distributed path on 1 GPU (RCCL all-gather, world_size 1)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --force-dist --exchange allgather --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_dist1_allgather.json 2>> $OUT/bench_dist1.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --force-dist --no-overlap --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_dist1_nooverlap.json 2>> $OUT/bench_dist1.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "%.3e vox/s"%j["value"], "%.2f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "frac", j["roofline"]["frac"], "pipe", j["roofline"]["pipeline_frac"], j.get("box_filter_pass"))
    except Exception as e:
        print(f, "ERR", e)
PY
echo "== rocprofv3 kernel trace (same command as the bench)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/rocprof_stdout.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fgf -o trace -- python $GRAFT_REPO_ROOT/bench.py --fgf 4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/rocprof_fgf_stdout.log 2>&1
echo "== rocprofv3 PMC passes (HBM traffic of the hot kernels)"
for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr:WRITE_SIZE TCC_EA0_WRREQ_sum" "ft:FETCH_SIZE" "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "lds:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  n=${pass%%:*}; c=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 0 > $OUT/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $GRAFT_REPO_ROOT
for n in rd wr ft sq lds; do f=$(find $OUT/pmc_$n -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f > $OUT/pmc_$n.summary.txt 2>&1; done
python scripts/make_traffic.py $OUT > $OUT/traffic.log 2>&1; tail -5 $OUT/traffic.log
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls $OUT/prof/* 2>/dev/null | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
echo "== done"
