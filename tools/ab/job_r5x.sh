#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5x
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_lift_gpu.py tests/test_k1_gpu.py -q -m gpu 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_modules_gpu.py -q -m gpu -k "cat128 or fullsize" 2>&1 | tail -4
for t16 in 1 0; do UBV_LIFT_TILE16=$t16 timeout 600 python bench.py --workload LC_cat128 --dtype fp32 --no-cpu-baseline --no-extras --no-ieee-gemm --params init --extras-file $OUT/cat128_tile16_$t16.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UBV_LIFT_TILE16=$t16', d['value'], d['ms_per_step'], d['config']['parity']); print(d['roofline_ops'])"; done
for t16 in 1 0; do echo "tile16=$t16"; UBV_LIFT_TILE16=$t16 timeout 300 python tools/bench_lift.py --dtype fp32 --img-hw 800 1440 --dh 16 --only self,pts 2>&1 | grep -E "^self|^pts"; done
