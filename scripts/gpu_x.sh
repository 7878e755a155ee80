#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stripes.py -m gpu -q -x -p no:cacheprovider --timeout=600 2>&1 | tail -15
