#!/bin/bash
# in-step durations of the f32 gemm_nt<4> launches grouped by grid size (= shape): eager single-stream steps under rocprofv3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg
timeout 300 rocprofv3 --kernel-trace -d /tmp/sg -o e -- python $ROOT/bench.py --dtype fp32 --no-graph --single-stream --no-extras --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 2 > /dev/null 2>&1
python $ROOT/tools/db_dispatches.py $(find /tmp/sg -name 'e_results.db') "gemm_nt_kernelILi4ELb1" 498 | awk '{d[$4]+=$1; n[$4]++; if (!($4 in mn) || $1<mn[$4]) mn[$4]=$1; if ($1>mx[$4]) mx[$4]=$1} END {for (g in d) printf "grid %9d  n %4d  avg %7.1f  min %7.1f  max %7.1f us\n", g, n[g], d[g]/n[g], mn[g], mx[g]}' | sort -k2 -n
python $ROOT/tools/db_dispatches.py $(find /tmp/sg -name 'e_results.db') "gemm_nt_kernelILi4ELb1" 170 | tail -84 | awk '{printf "%s:%s ", $1, $4} END {print ""}'
