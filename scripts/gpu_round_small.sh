#!/bin/bash
# Part of a round's GPU session again, after a change that only touches the Middlebury-size configurations (c2, c1, c1x: the narrow
# layout and the flow cost model of round 5): smoke, their bench lines (single pair, fixtures, frames in flight, batches) and the
# PMC passes of c2, written over the files of the full session (scripts/gpu_round.sh) in gpurun_out/<tag>; device_small.txt names
# the box.  Usage (via gpurun): bash scripts/gpu_round_small.sh r05
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
GFX=$(rocminfo 2>/dev/null | grep -m1 -oE "gfx9[0-9a-f]+")
CPU=$(grep -m1 "model name" /proc/cpuinfo | sed 's/.*: //')
echo "$TAG (Middlebury-size lines re-measured) $(date -u +%Y-%m-%dT%H:%MZ): $GFX (MI355X), host $CPU, $(nproc) cores, box $(hostname)" > $OUT/device_small.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
cd /tmp
for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "wr:WRITE_SIZE TCC_EA0_WRREQ_sum" "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=${pass%%:*}; c=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_c2_$n -o $n -- python $GRAFT_REPO_ROOT/scripts/prof_run.py c2 > $OUT/pmc_c2_$n.log 2>&1 || echo "pmc pass c2 $n failed"
  fdb=$(find $OUT/pmc_c2_$n -name "*.db" | head -1); [ -n "$fdb" ] && python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $fdb > $OUT/pmc_c2_$n.summary.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/make_traffic.py $OUT $TAG > $OUT/traffic_small.log 2>&1; grep -c "keeping the committed" $OUT/traffic_small.log
cp profiles/traffic.json $OUT/traffic.json
B="timeout 600 python bench.py"
$B --config c2 --steps 30 --verify --pp > $OUT/bench_c2_n1.json 2>> $OUT/bench_small.err
$B --config c1 --steps 30 --verify > $OUT/bench_c1_u8_n1.json 2>> $OUT/bench_small.err
$B --config c1x --steps 30 --verify > $OUT/bench_c1x_u8_n1.json 2>> $OUT/bench_small.err
$B --config c2 --pair fixture --steps 30 --verify --no-cpu-wide > $OUT/bench_c2_teddy_n1.json 2>> $OUT/bench_small.err
$B --config c1 --pair fixture --steps 30 --verify --no-cpu-wide > $OUT/bench_c1_cones_u8_n1.json 2>> $OUT/bench_small.err
$B --config c1x --pair fixture --steps 30 --verify --no-cpu-wide > $OUT/bench_c1x_cones_u8_n1.json 2>> $OUT/bench_small.err
for cfg in c2 c1 c1x; do $B --config $cfg --frames-in-flight 2 --steps 40 --verify --no-cpu-wide > $OUT/bench_${cfg}_fif2.json 2>> $OUT/bench_small.err; done
$B --config c2 --pair fixture --frames-in-flight 2 --steps 40 --verify --no-cpu-wide > $OUT/bench_c2_teddy_fif2.json 2>> $OUT/bench_small.err
$B --config c1 --pair fixture --frames-in-flight 2 --steps 40 --verify --no-cpu-wide > $OUT/bench_c1_cones_u8_fif2.json 2>> $OUT/bench_small.err
$B --config c2 --pair fixture --frames-in-flight 2 --seg-rows 375 --steps 40 --verify --no-cpu-wide > $OUT/bench_c2_teddy_fif2_seg375.json 2>> $OUT/bench_small.err
$B --config c1 --pair fixture --frames-in-flight 2 --seg-rows 375 --steps 40 --verify --no-cpu-wide > $OUT/bench_c1_cones_u8_fif2_seg375.json 2>> $OUT/bench_small.err
$B --config c1x --pair fixture --frames-in-flight 2 --seg-rows 288 --steps 40 --verify --no-cpu-wide > $OUT/bench_c1x_cones_u8_fif2_seg288.json 2>> $OUT/bench_small.err
for cfg in c2 c1 c1x; do for b in 2 4 8 16; do $B --config $cfg --batch $b --steps 30 --warmup 5 --no-cpu-wide > $OUT/bench_${cfg}_batch$b.json 2>> $OUT/bench_small.err; done; done
$B --config c2 --batch 8 --graph --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_batch8_graph.json 2>> $OUT/bench_small.err
$B --config c2 --batch -1 --graph --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_batch1_graph.json 2>> $OUT/bench_small.err
# the headline and c3 on this box as well (unchanged code path: a box-to-box reference for the lines above)
$B --steps 10 --warmup 3 --no-cpu-baseline --frame-loop 0 > $OUT/bench_c4_small_session_ref.json 2>> $OUT/bench_small.err
$B --config c3 --no-cpu-baseline > $OUT/bench_c3_small_session_ref.json 2>> $OUT/bench_small.err
tail -3 $OUT/bench_small.err
python - <<PY
import json,glob,os,time
for f in sorted(glob.glob("$OUT/bench_*.json")):
    if time.time() - os.path.getmtime(f) > 3000: continue
    try:
        j=json.loads([ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")][-1])
        print(f.split('/')[-1], "%.3e vox/s"%j["value"], "%.4f ms"%j["ms_per_step"], "frac", j["roofline"]["frac"], "binding", j["roofline"].get("binding_frac"), "oracle", j.get("oracle_maps_equal"), "cpu", (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
find $OUT -name "*.db" -delete
echo "== done"
