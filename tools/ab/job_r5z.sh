#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5z
mkdir -p $OUT
cd $ROOT
UBV_TRACE_MIN=1000 timeout 600 python tools/trace_elementwise.py 2>&1 | grep -v "^/opt" > $OUT/trace_small.txt
wc -l $OUT/trace_small.txt; head -70 $OUT/trace_small.txt
