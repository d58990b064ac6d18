#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5a4; mkdir -p $O
PREV=$PWD/unibev_amd/libunibev_hip_prev.so
timeout 1700 python -m pytest tests/test_gemm_gpu.py tests/test_sparse_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for lib in prev new prev new; do
  echo "== $lib" | tee -a $O/ab.txt
  if [ $lib = prev ]; then export UBV_LIB_PATH=$PREV; else unset UBV_LIB_PATH; fi
  TAG=hot timeout 300 python tools/ab/wgrad_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
