#!/bin/bash
# FGF row on the GPU box: parity tests + per-kernel timing of the FGF pipeline at full HD.
OUT=gpurun_out/${1:-fgf}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo "BUILD FAILED"; tail -20 $OUT/build.log; }
timeout 900 python -m pytest tests/test_gpu_fgf.py -m gpu -q -x -s -p no:cacheprovider --timeout=600 > $OUT/pytest_fgf.log 2>&1
grep -E "parity|passed|failed|Error|error" $OUT/pytest_fgf.log | tail -40
for s in 2 4 8; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --fgf $s > $OUT/bench_fgf_s$s.json 2> $OUT/bench_fgf_s$s.err || tail -3 $OUT/bench_fgf_s$s.err
  tail -1 $OUT/bench_fgf_s$s.json
done
