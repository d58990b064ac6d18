#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1500 python -m pytest tests/test_bench_gpu.py -q -m gpu -k "overlapped or contract" 2>&1 | tail -12
