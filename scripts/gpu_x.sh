#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pp_ocv.py -m gpu -q -x -p no:cacheprovider --timeout=300 2>&1 | tail -2
timeout 300 python scripts/dbg_wmf.py big 2>&1 | tail -5
