#!/bin/bash
# On the GPU box: PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only) per sampling op,
# then tools/make_traffic.py -> $OUT/traffic.json
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/traffic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dt in bf16 fp32; do
for op in self pts img; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/traffic_db -o ${op}_${dt}_${ctr} -- python $ROOT/tools/bench_lift.py --only $op --iters 3 --dtype $dt > /dev/null 2>&1
  done
done
done
python $ROOT/tools/make_traffic.py /tmp/traffic_db > $OUT/traffic.json
