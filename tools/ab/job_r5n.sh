#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py"
for ws in 1 0; do
  UBV_GEMM_WS=$ws rocprofv3 --kernel-trace -d /tmp/prof_ws$ws -o e -- $B --dtype fp32 --no-graph --single-stream --no-extras --no-cpu-baseline --no-kernel-timing --no-parity --params init --no-ieee-gemm --steps 10 --warmup 3 --extras-file '' > $OUT/bench_eager_ws$ws.json 2>/dev/null
  python $ROOT/tools/db_table.py /tmp/prof_ws$ws/e_results.db 24 40 > $OUT/fp32_eager_kernel_table_ws$ws.txt
  head -24 $OUT/fp32_eager_kernel_table_ws$ws.txt
done
for ws in 1 0; do UBV_GEMM_WS=$ws timeout 600 python $ROOT/bench.py --dtype fp32 --no-extras --no-cpu-baseline --params init --no-ieee-gemm --no-kernel-timing --no-parity --single-stream --extras-file '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one stream UBV_GEMM_WS=$ws', d['value'], d['ms_per_step'])"; done
