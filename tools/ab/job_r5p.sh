#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5p
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_modules_gpu.py -q -m gpu -k "fullsize_gradients_vs_oracle" 2>&1 | tail -15 > $OUT/grad_tests.txt; cat $OUT/grad_tests.txt
timeout 1500 python -m pytest tests/test_bench_gpu.py -q -m gpu 2>&1 | tail -15 > $OUT/bench_tests.txt; cat $OUT/bench_tests.txt
