#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5w4; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "wgrad" > $O/tests_gemm.txt 2>&1; tail -2 $O/tests_gemm.txt
UBV_WGRAD_PW=4 timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "wgrad" > $O/tests_gemm4.txt 2>&1; tail -2 $O/tests_gemm4.txt
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu > $O/tests_sparse.txt 2>&1; tail -2 $O/tests_sparse.txt
UBV_WGRAD_PW=4 timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu > $O/tests_sparse4.txt 2>&1; tail -2 $O/tests_sparse4.txt
run() { echo "== $*" | tee -a $O/ab.txt
  env "$@" TAG=hot timeout 300 python tools/ab/wgrad_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  env "$@" timeout 300 python tools/bench_gemm_cold.py 2>&1 | tail -4 | cut -c60- | tee -a $O/ab.txt
  env "$@" UBV_KEEP_RULEBOOKS=1 TAG=kept timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
}
run UBV_WGRAD_WS=0
run UBV_WGRAD_PW=4
run UBV_WGRAD_PW=8
for b in 256 512 1024; do
  for pw in 4 8; do
    echo "== spconv blocks $b pw $pw" | tee -a $O/ab.txt
    UBV_SPCONV_WGRAD_BLOCKS=$b UBV_WGRAD_PW=$pw UBV_KEEP_RULEBOOKS=1 TAG=kept timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  done
done
