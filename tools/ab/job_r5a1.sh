#!/bin/bash
# 4-wave weight-gradient kernel with 32-bit row arithmetic + unmasked fast path against the previous build (UBV_LIB_PATH)
export TMPDIR=/tmp
O=gpurun_out/r5a1; mkdir -p $O
PREV=$PWD/unibev_amd/libunibev_hip_prev.so
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_sparse_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for lib in prev new prev new; do
  echo "== $lib" | tee -a $O/ab.txt
  if [ $lib = prev ]; then export UBV_LIB_PATH=$PREV; else unset UBV_LIB_PATH; fi
  TAG=hot timeout 300 python tools/ab/wgrad_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  timeout 300 python tools/bench_gemm_cold.py 2>&1 | tail -4 | cut -c60-130 | tee -a $O/ab.txt
  UBV_SPCONV_WGRAD_WS=0 UBV_KEEP_RULEBOOKS=1 TAG=kept-4wave timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
