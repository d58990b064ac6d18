#!/bin/bash
# On the GPU box: per-kernel durations of the lifting micro-benchmark (rocprofv3 kernel trace).
#   bash tools/prof_lift.sh <tag> [bench_lift.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o t -- python $ROOT/tools/bench_lift.py "$@" > /tmp/prof_$TAG.log 2>&1
python $ROOT/tools/db_table.py $(find /tmp/prof_$TAG -name '*_results.db' | head -1) 1 40 | grep -v "at::\|elementwise\|Rand\|distribution" | cut -c1-150
