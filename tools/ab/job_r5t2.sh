#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5t2; mkdir -p $O
timeout 1700 python -m pytest tests/test_lift_gpu.py tests/test_modules_gpu.py tests/test_k1_gpu.py tests/test_bench_gpu.py -q -m gpu > $O/tests.txt 2>&1; tail -6 $O/tests.txt
UBV_LIFT_TILE_LP=2 timeout 900 python -m pytest tests/test_lift_gpu.py -q -m gpu > $O/tests_lp2.txt 2>&1; tail -4 $O/tests_lp2.txt
for lp in 0 1; do
  echo "== UBV_LIFT_TILE_LP=$lp" | tee -a $O/bench.txt
  UBV_LIFT_TILE_LP=$lp timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['lowp'])" | tee -a $O/bench.txt
done
