#!/bin/bash
# SQ counters of the weight-gradient kernel (two PMC passes): tools/pmc_wgrad.sh <outfile>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/pmc_wgrad.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_wg
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_wg -o a -- python $ROOT/tools/ab/wgrad_only.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_wg -o b -- python $ROOT/tools/ab/wgrad_only.py > /dev/null 2>&1
for f in $(find /tmp/pmc_wg -name '*_results.db'); do echo "== $f"; python $ROOT/tools/pmc_db.py $f gemm_wgrad; done > $OUT 2>&1
