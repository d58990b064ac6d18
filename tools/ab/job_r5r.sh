#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for cap in 0 224 192 128; do UBV_WS_CUS=$cap timeout 600 python bench.py --dtype fp32 --no-extras --no-cpu-baseline --params init --no-ieee-gemm --no-kernel-timing --no-parity --extras-file '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two streams UBV_WS_CUS=$cap', d['value'], d['ms_per_step'])"; done
