#!/bin/bash
# SQ / cache counters of the lifting kernels matching <pattern> (three PMC passes, kernel trace only):
#   bash tools/pmc_lift.sh <outfile> <kernel-name pattern> [bench_lift.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; PAT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lift
B="python $ROOT/tools/bench_lift.py --iters 3 --dtype fp32 $@"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_lift -o a -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_lift -o b -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d /tmp/pmc_lift -o c -- $B > /dev/null 2>&1
for f in $(find /tmp/pmc_lift -name '*_results.db' | sort); do echo "== $f"; python $ROOT/tools/pmc_db.py $f "$PAT"; done > $OUT 2>&1
