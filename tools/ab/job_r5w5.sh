#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5w5; mkdir -p $O
for cfg in "UBV_WGRAD_WS=0" "UBV_WGRAD_PW=8" "UBV_WGRAD_PW=4" "UBV_WGRAD_WS=0" "UBV_WGRAD_PW=8"; do
  echo "== $cfg" | tee -a $O/bench.txt
  env $cfg timeout 600 python bench.py --no-cpu-baseline --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('voxel'))" | tee -a $O/bench.txt
done
