#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5c6; mkdir -p $O
for cfg in "X=0" "UBV_WGRAD_BLOCKS=256" "UBV_WGRAD_BLOCKS=384" "UBV_WGRAD_WS=1" "X=0"; do
  echo "== $cfg" | tee -a $O/bench.txt
  env $cfg timeout 600 python bench.py --workload LC_cat128 --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done
