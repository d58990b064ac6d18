#!/bin/bash
# SQ / cache counters of the f32 gemm_nt kernel (separate PMC passes, each under its own timeout):
#   tools/pmc_gemm.sh <outfile> [kernel substring]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/${1:-gpurun_out/pmc_gemm.txt}; mkdir -p $(dirname $OUT)
SUB=${2:-gemm_nt}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_g
P="python $ROOT/tools/ab/gemm_one.py"
: > $OUT
pass() {  # name counters...
  local name=$1; shift
  timeout 100 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_g -o $name -- $P > /tmp/pmc_g_$name.log 2>&1
  echo "== pass $name rc=$?" >> $OUT
  for f in $(find /tmp/pmc_g -name "${name}_results.db"); do python $ROOT/tools/pmc_db.py $f $SUB >> $OUT 2>&1; done
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
pass b SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass c SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass d TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum
