#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5w8; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_sparse_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for m in 1 2 4 8; do
  for ws in 0 8; do
    echo "== mult $m  sparse ws $ws" | tee -a $O/ab.txt
    UBV_SPWG_MULT=$m UBV_SPCONV_WGRAD_WS=$ws UBV_KEEP_RULEBOOKS=1 TAG=kept timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
    UBV_SPWG_MULT=$m UBV_SPCONV_WGRAD_WS=$ws UBV_KEEP_RULEBOOKS=0 TAG=rebuilt timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  done
done
