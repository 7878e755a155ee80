#!/bin/bash
# Copies the artifacts of one scripts/gpu_round.sh session (gpurun_out/<tag>) into the tracked profiles/<tag>/.
# Usage: bash scripts/collect_profiles.sh r03
TAG=${1:-r06}
S=gpurun_out/$TAG
D=profiles/$TAG
mkdir -p $D
cp $S/bench_*.json $S/device.txt $S/device_small.txt $S/smoke.log $S/trace_gaps_c4.txt $S/trace_gaps_shard8.txt $S/wmf_timing.txt $S/soak.txt $S/pytest_gpu_reports.txt $D/ 2>/dev/null
tail -4 $S/pytest_gpu.log > $D/pytest_gpu_tail.txt
cp $(find $S/prof -name "*kernel_stats.csv" | head -1) $D/rocprofv3_kernel_stats_bench_c4.csv
cp $(find $S/prof_s8 -name "*kernel_stats.csv" | head -1) $D/rocprofv3_kernel_stats_bench_c4_shardsim8.csv
for n in rd wr ft sq sqb lds tcc; do cp $S/pmc_$n.summary.txt $D/rocprofv3_pmc_$n.summary.txt; done
for n in rd wr sq; do cp $S/pmc_u8_$n.summary.txt $D/rocprofv3_pmc_u8_$n.summary.txt; done
for c in c3 c2 c5; do for n in rd wr sq tcc; do [ -f $S/pmc_${c}_$n.summary.txt ] && cp $S/pmc_${c}_$n.summary.txt $D/rocprofv3_pmc_${c}_$n.summary.txt; done; done
[ -f $S/isa_mix.json ] && cp $S/isa_mix.json $D/isa_mix.json
cp $S/bench_world2.err $D/bench_world2_stderr.txt 2>/dev/null
for v in exact tol; do [ -f $S/pmc_sq_$v.summary.txt ] && cp $S/pmc_sq_$v.summary.txt $D/rocprofv3_pmc_sq_$v.summary.txt; done
cp $S/traffic.json profiles/traffic.json
ls $D | wc -l
