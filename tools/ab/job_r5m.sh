#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5m
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gemm_gpu.py > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt
for ws in 1 0; do UBV_GEMM_WS=$ws timeout 600 python bench.py --dtype fp32 --no-extras --no-cpu-baseline --params init --no-ieee-gemm --no-kernel-timing --extras-file $OUT/extras_ws$ws.json > $OUT/bench_ws$ws.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_ws$ws.json')); print('UBV_GEMM_WS=$ws', d['value'], d['ms_per_step'], d['config']['parity'])"; done
