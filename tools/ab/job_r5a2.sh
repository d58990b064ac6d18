#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5a2; mkdir -p $O
for rep in 1 2; do
for ws in 0 8; do
  for keep in 1 0; do
    UBV_SPCONV_WGRAD_WS=$ws UBV_KEEP_RULEBOOKS=$keep TAG="ws=$ws keep=$keep" timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  done
done
done
for cfg in "X=0" "UBV_WGRAD_WS=1" "X=0" "UBV_WGRAD_WS=1"; do
  echo "== step $cfg" | tee -a $O/ab.txt
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
