#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pp_ocv.py -m gpu -q -x -s -p no:cacheprovider --timeout=300 -k "wgt_median" 2>&1 | grep -E "wmf|passed|failed|Error|error" | tail -40
timeout 300 python scripts/dbg_wmf.py
