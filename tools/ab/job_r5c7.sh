#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5c7; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k wgrad 2>&1 | tail -2
for cfg in "X=0" "UBV_WGRAD_BLOCKS=384" "UBV_WGRAD_BLOCKS=640" "UBV_WGRAD_BLOCKS=768" "X=0"; do
  echo "== $cfg" | tee -a $O/bench.txt
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done
echo "== cat128 default" | tee -a $O/bench.txt
timeout 600 python bench.py --workload LC_cat128 --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
