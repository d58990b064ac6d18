#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5w7; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_sparse_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
