#!/bin/bash
# TILE plan: largest pixel box a block copies into LDS (UBV_TILE_MAXBOX_FWD / _BWD), spread and init operating points
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5o
mkdir -p $OUT
cd $ROOT
for mode in "--random-offsets" ""; do
for mb in 256 196 144 100 64 0; do
  echo "== offsets: ${mode:-init}  max box $mb"
  UBV_TILE_MAXBOX_FWD=$mb UBV_TILE_MAXBOX_BWD=$mb timeout 300 python tools/bench_lift.py --dtype fp32 --only self,pts $mode 2>&1 | grep -E "lift_tile|^self|^pts|bwd_value"
done; done > $OUT/tile_maxbox.txt 2>&1
cat $OUT/tile_maxbox.txt
