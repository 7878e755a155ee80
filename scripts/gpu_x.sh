#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/dbg_stripe.py 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_stripes.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 2>&1 | tail -4
