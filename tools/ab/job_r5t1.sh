#!/bin/bash
# TILE plan for 16-bit value maps: parity tests, then operator timings against the window / gather kernels (UBV_LIFT_TILE_LP=0)
export TMPDIR=/tmp
O=gpurun_out/r5t1; mkdir -p $O
timeout 1500 python -m pytest tests/test_lift_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for lp in 0 1; do
  for dt in bf16 fp16; do
    echo "== UBV_LIFT_TILE_LP=$lp $dt" | tee -a $O/lift.txt
    UBV_LIFT_TILE_LP=$lp timeout 300 python tools/bench_lift.py --dtype $dt 2>&1 | grep -v "^/opt" | grep "B=2 $dt\|bwd_op\|lift_fwd" | tee -a $O/lift.txt
  done
done
