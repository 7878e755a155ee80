#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=600 2>&1 | tail -5
