#!/bin/bash
# Copies the artifacts of one scripts/gpu_round.sh session (gpurun_out/<tag>) into the tracked profiles/<tag>/.
# Usage: bash scripts/collect_profiles.sh r02
TAG=${1:-r02}
S=gpurun_out/$TAG
D=profiles/$TAG
mkdir -p $D
cp $S/bench_*.json $S/device.txt $S/smoke.log $S/role_cycles.txt $S/trace_gaps_c4.txt $S/trace_gaps_shard8.txt $S/wmf_timing.txt $S/pytest_gpu_reports.txt $D/ 2>/dev/null
tail -4 $S/pytest_gpu.log > $D/pytest_gpu_tail.txt
cp $S/prof/trace_kernel_stats.csv $D/rocprofv3_kernel_stats_bench_c4.csv
cp $S/prof_s8/trace_kernel_stats.csv $D/rocprofv3_kernel_stats_bench_c4_shardsim8.csv
for n in rd wr ft sq sqb lds; do cp $S/pmc_$n.summary.txt $D/rocprofv3_pmc_$n.summary.txt; done
for n in sq lds; do cp $S/pmcq2_$n.summary.txt $D/rocprofv3_pmc_q2_variant_$n.summary.txt; done
cp $S/traffic.json profiles/traffic.json
ls $D | wc -l
