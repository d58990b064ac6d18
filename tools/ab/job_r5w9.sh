#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5w9; mkdir -p $O
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_glue_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done
