#!/bin/bash
# step-level A/B of the wave-specialised weight gradient: two streams (default) and one stream
export TMPDIR=/tmp
O=gpurun_out/r5w6; mkdir -p $O
for rep in 1 2; do
for cfg in "UBV_WGRAD_WS=0" "UBV_WGRAD_PW=8" "UBV_WGRAD_PW=4"; do
  for ss in "" "--single-stream"; do
    echo "== $cfg $ss" | tee -a $O/bench.txt
    env $cfg timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' $ss 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
  done
done
done
