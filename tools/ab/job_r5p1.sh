#!/bin/bash
# two-stream scheduling knobs: side-stream priority, which encoder takes the side stream
export TMPDIR=/tmp
O=gpurun_out/r5p1; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" | tee -a $O/bench.txt
for rep in 1 2; do
for cfg in "X=0" "UBV_SIDE_PRIORITY=-1" "UBV_SIDE_IS_IMG=1" "UBV_SIDE_IS_IMG=1 UBV_SIDE_PRIORITY=-1"; do
  echo "== $cfg" | tee -a $O/bench.txt
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-extras --extras-file '' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done
done
