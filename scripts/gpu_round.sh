#!/bin/bash
# One full GPU-box session of a round: parity tests, smoke, benches of every config, the torch.distributed path on
# one GPU, rocprofv3 kernel trace + PMC passes (each in its own run).  Usage (via gpurun): bash scripts/gpu_round.sh r03
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
GFX=$(rocminfo 2>/dev/null | grep -m1 -oE "gfx9[0-9a-f]+")
CPU=$(grep -m1 "model name" /proc/cpuinfo | sed 's/.*: //')
echo "$TAG $(date -u +%Y-%m-%dT%H:%MZ): $GFX (MI355X), host $CPU, $(nproc) cores, box $(hostname)" > $OUT/device.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout=1200 > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
grep -E "^\.?\[wmf\]|^\.?\[ocv-order\]|^\.?\[tol\]" $OUT/pytest_gpu.log > $OUT/pytest_gpu_reports.txt
echo "== PMC passes first (own runs; kernel trace only): the headline line below then carries THIS session's traffic figure"
cd /tmp
for dt in f32 u8; do
  pre=pmc_; [ $dt = u8 ] && pre=pmc_u8_
  for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr:WRITE_SIZE TCC_EA0_WRREQ_sum" "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    n=${pass%%:*}; c=${pass#*:}
    PSM_DTYPE=$dt timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/${pre}$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 > $OUT/${pre}$n.log 2>&1 || echo "pmc pass $dt $n failed"
    fdb=$(find $OUT/${pre}$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/${pre}$n.summary.txt 2>&1
  done
done
for pass in "ft:FETCH_SIZE" "sqb:SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" "lds:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" "tcc:TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=${pass%%:*}; c=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 > $OUT/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
  fdb=$(find $OUT/pmc_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_$n.summary.txt 2>&1
done
for v in exact tol; do
  fl=0; [ $v = tol ] && fl=33554432
  PSM_FLAGS=$fl timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES -d $OUT/pmc_sq_$v -o sq -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c4 > $OUT/pmc_sq_$v.log 2>&1 || echo "pmc $v failed"
  fdb=$(find $OUT/pmc_sq_$v -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_sq_$v.summary.txt 2>&1
done
echo "== PMC passes for BASELINE configs[2] (c3: 1280x720x128), c2 and c5 as well"
for cfg in c3 c2 c5; do
  for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr:WRITE_SIZE TCC_EA0_WRREQ_sum" "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "tcc:TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=${pass%%:*}; c=${pass#*:}
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${cfg}_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py $cfg > $OUT/pmc_${cfg}_$n.log 2>&1 || echo "pmc pass $cfg $n failed"
    fdb=$(find $OUT/pmc_${cfg}_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_${cfg}_$n.summary.txt 2>&1
  done
done
cd $GRAFT_REPO_ROOT
mkdir -p profiles/$TAG; python scripts/isa_mix.py --json profiles/$TAG/isa_mix.json > $OUT/isa_mix.log 2>&1; cp profiles/$TAG/isa_mix.json $OUT/isa_mix.json
python scripts/make_traffic.py $OUT $TAG > $OUT/traffic.log 2>&1; tail -12 $OUT/traffic.log
cp profiles/traffic.json $OUT/traffic.json
echo "== bench (default: c4, N=1) - the headline line"
timeout 900 python bench.py --box-bench --verify --pp > $OUT/bench_c4_n1.json 2> $OUT/bench_c4.err; cat $OUT/bench_c4_n1.json; tail -2 $OUT/bench_c4.err
B="timeout 600 python bench.py"
$B --config c3 --steps 30 --warmup 5 --verify --pp > $OUT/bench_c3_n1.json 2>> $OUT/bench_var.err
$B --config c2 --steps 30 --verify --pp > $OUT/bench_c2_n1.json 2>> $OUT/bench_var.err
$B --config c5 --steps 4 --warmup 1 --verify --no-cpu-wide > $OUT/bench_c5_n1.json 2>> $OUT/bench_var.err
$B --config c1 --steps 30 --verify > $OUT/bench_c1_u8_n1.json 2>> $OUT/bench_var.err
$B --config c1x --steps 30 --verify > $OUT/bench_c1x_u8_n1.json 2>> $OUT/bench_var.err
echo "== the Middlebury pairs BASELINE configs[0] / [1] name (tests/golden fixtures) and two frames in flight on the configurations below the headline"
$B --config c2 --pair fixture --steps 30 --verify --no-cpu-wide > $OUT/bench_c2_teddy_n1.json 2>> $OUT/bench_var.err
$B --config c1 --pair fixture --steps 30 --verify --no-cpu-wide > $OUT/bench_c1_cones_u8_n1.json 2>> $OUT/bench_var.err
$B --config c1x --pair fixture --steps 30 --verify --no-cpu-wide > $OUT/bench_c1x_cones_u8_n1.json 2>> $OUT/bench_var.err
for cfg in c3 c2 c1 c1x; do $B --config $cfg --frames-in-flight 2 --steps 40 --verify --no-cpu-wide > $OUT/bench_${cfg}_fif2.json 2>> $OUT/bench_var.err; done
$B --config c2 --pair fixture --frames-in-flight 2 --steps 40 --verify --no-cpu-wide > $OUT/bench_c2_teddy_fif2.json 2>> $OUT/bench_var.err
$B --config c1 --pair fixture --frames-in-flight 2 --steps 40 --verify --no-cpu-wide > $OUT/bench_c1_cones_u8_fif2.json 2>> $OUT/bench_var.err
# (ONE segment per launch was the round's first answer for a small launch with a second frame filling its tail; the ring's contexts now
#  carry PSM_OPT_FRAMES_IN_FLIGHT and the planner's own cut is the faster one - these lines stay as the comparison)
$B --config c2 --pair fixture --frames-in-flight 2 --seg-rows 375 --steps 40 --verify --no-cpu-wide > $OUT/bench_c2_teddy_fif2_seg375.json 2>> $OUT/bench_var.err
$B --config c1 --pair fixture --frames-in-flight 2 --seg-rows 375 --steps 40 --verify --no-cpu-wide > $OUT/bench_c1_cones_u8_fif2_seg375.json 2>> $OUT/bench_var.err
$B --config c1x --pair fixture --frames-in-flight 2 --seg-rows 288 --steps 40 --verify --no-cpu-wide > $OUT/bench_c1x_cones_u8_fif2_seg288.json 2>> $OUT/bench_var.err
$B --frames-in-flight 2 --no-cpu-baseline --frame-loop 0 > $OUT/bench_c4_fif2.json 2>> $OUT/bench_var.err
$B --flags 67108864 --no-cpu-wide --frame-loop 0 > $OUT/bench_c4_fma_solve.json 2>> $OUT/bench_var.err
$B --config c4 --dtype u8 --verify > $OUT/bench_c4_u8_n1.json 2>> $OUT/bench_var.err
$B --flags 2097152 --no-cpu-baseline --verify > $OUT/bench_c4_single_phase_n1.json 2>> $OUT/bench_var.err
$B --flags 8192 --no-cpu-baseline --verify > $OUT/bench_c4_store_mode_n1.json 2>> $OUT/bench_var.err
for s in 2 4 8; do $B --fgf $s --no-cpu-wide > $OUT/bench_c4_fgf_s${s}_n1.json 2>> $OUT/bench_var.err; done
echo "== tolerance form (PSM_FLAG_F32_TOL), same box, alternating with the default"
for rep in 1 2; do
  $B --steps 20 --warmup 5 --no-cpu-baseline --frame-loop 0 > $OUT/bench_c4_exact_ab$rep.json 2>> $OUT/bench_var.err
  $B --steps 20 --warmup 5 --flags 33554432 --no-cpu-wide --frame-loop 0 > $OUT/bench_c4_tol_ab$rep.json 2>> $OUT/bench_var.err
done
echo "== batches of Middlebury-size pairs (psm_compute_batch)"
for cfg in c2 c1 c1x; do for b in 2 4 8 16; do $B --config $cfg --batch $b --steps 30 --warmup 5 --no-cpu-wide > $OUT/bench_${cfg}_batch$b.json 2>> $OUT/bench_var.err; done; done
for g in 2 4 8; do $B --shard-sim $g --steps 40 > $OUT/bench_c4_shardsim_1of$g.json 2>> $OUT/bench_var.err; done
for g in 2 4 8; do $B --shard-sim $g --shard disp --steps 40 > $OUT/bench_c4_shardsim_disp_1of$g.json 2>> $OUT/bench_var.err; done
$B --shard-sim 8 --frames-in-flight 2 --steps 40 > $OUT/bench_c4_shardsim_1of8_fif2.json 2>> $OUT/bench_var.err
$B --shard-sim 8 --shard disp --frames-in-flight 2 --steps 40 > $OUT/bench_c4_shardsim_disp_1of8_fif2.json 2>> $OUT/bench_var.err
$B --shard-sim 8 --shard disp --strided --steps 40 > $OUT/bench_c4_shardsim_disp_strided_1of8.json 2>> $OUT/bench_var.err
$B --config c5 --shard-sim 8 --steps 6 --warmup 2 > $OUT/bench_c5_shardsim_1of8.json 2>> $OUT/bench_var.err
echo "== weighted median timing (hybrid sweeps form / dataflow form)"
timeout 300 python scripts/dbg_wmf.py big > $OUT/wmf_timing.txt 2>&1; WM_FLAGS=4194304 timeout 300 python scripts/dbg_wmf.py >> $OUT/wmf_timing.txt 2>&1; tail -22 $OUT/wmf_timing.txt
echo "== torch.distributed path on 1 GPU (RCCL, world_size 1): both sharding axes per invocation"
D="timeout 600 python bench.py --gpus 1 --force-dist --frames-in-flight 2 --steps 10 --warmup 3 --no-cpu-baseline"   # (two frames in flight per rank: what --gpus N runs from N = 4)
$D > $OUT/bench_c4_dist_world1.json 2> $OUT/bench_dist1.err
$D --shard disp > $OUT/bench_c4_dist_world1_disp.json 2>> $OUT/bench_dist1.err
$D --no-frame-pipeline > $OUT/bench_c4_dist_world1_nopipeline.json 2>> $OUT/bench_dist1.err
${D/--frames-in-flight 2/--frames-in-flight 1} > $OUT/bench_c4_dist_world1_fif1.json 2>> $OUT/bench_dist1.err
echo "== the N > 1 protocol with TWO ranks on this one GPU (RCCL first; gloo-staged exchange when RCCL refuses the duplicate device)"
W2="timeout 900 python bench.py --gpus 2 --same-device --frames-in-flight 2 --steps 10 --warmup 3"
$W2 > $OUT/bench_c4_world2_same_device.json 2> $OUT/bench_world2.err
$W2 --shard disp > $OUT/bench_c4_world2_same_device_disp.json 2>> $OUT/bench_world2.err
$W2 --no-frame-pipeline > $OUT/bench_c4_world2_same_device_nopipeline.json 2>> $OUT/bench_world2.err
grep -h "RCCL with" $OUT/bench_world2.err | head -3
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.3e vox/s"%j["value"], "%.3f ms"%j["ms_per_step"], {k:v["avg_ms"] for k,v in j["kernels"].items()}, "frac", j["roofline"]["frac"], "verified", j.get("verified_vs_single_gpu"), "oracle", j.get("oracle_maps_equal"), "cpu", (j.get("cpu_baseline") or {}).get("value"), "alt", (j.get("alt_shard") or {}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
echo "== randomised soaks of the C ABI (batches included) and of the context state machine against the oracle"
timeout 300 python scripts/soak.py 150 $RANDOM > $OUT/soak.txt 2>&1; timeout 200 python scripts/soak_state.py 90 $RANDOM >> $OUT/soak.txt 2>&1; tail -3 $OUT/soak.txt
echo "== rocprofv3 kernel trace (same command as the bench)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --frame-loop 0 > $OUT/rocprof_stdout.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f 20 > $OUT/trace_gaps_c4.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_s8 -o trace -- python $GRAFT_REPO_ROOT/bench.py --shard-sim 8 --steps 40 --warmup 3 --no-cpu-baseline > $OUT/rocprof_s8_stdout.log 2>&1
f=$(find $OUT/prof_s8 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f 20 > $OUT/trace_gaps_shard8.txt 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +3M -delete
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
echo "== done"
