#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_k1_gpu.py -q -m gpu -k "query_grid" 2>&1 | grep -E "^E|passed|failed" | head -30
