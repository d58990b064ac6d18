#!/bin/bash
# wave-specialised weight gradient: parity tests, then A/B against the 4-wave kernel (UBV_WGRAD_WS=0)
export TMPDIR=/tmp
O=gpurun_out/r5w1; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "wgrad" > $O/tests_gemm.txt 2>&1; tail -5 $O/tests_gemm.txt
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu > $O/tests_sparse.txt 2>&1; tail -5 $O/tests_sparse.txt
for ws in 0 1; do
  echo "== UBV_WGRAD_WS=$ws" | tee -a $O/ab.txt
  UBV_WGRAD_WS=$ws TAG=hot timeout 300 python tools/ab/wgrad_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  UBV_WGRAD_WS=$ws timeout 300 python tools/bench_gemm_cold.py 2>&1 | tail -4 | tee -a $O/ab.txt
  UBV_WGRAD_WS=$ws UBV_KEEP_RULEBOOKS=1 TAG=kept timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
done
