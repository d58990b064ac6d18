#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5c8; mkdir -p $O
for rep in 1 2; do
for cfg in "UBV_WGRAD_BLOCKS=512" "X=0"; do
  echo "== cat128 $cfg" | tee -a $O/bench.txt
  env $cfg timeout 600 python bench.py --workload LC_cat128 --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done
done
