#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "memory_latency" 2>&1 | tail -4
