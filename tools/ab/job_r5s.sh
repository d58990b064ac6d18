#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "wgrad" 2>&1 | tail -2
for lib in libunibev_hip.so libunibev_hip_nospread.so; do echo "== $lib"; UBV_LIB_PATH=$ROOT/unibev_amd/$lib timeout 300 python tools/bench_gemm_cold.py 2>&1 | grep "^N="; done
