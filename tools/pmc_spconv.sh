#!/bin/bash
# SQ / cache counters of one sparse convolution or its weight gradient (tools/ab/spconv_one.py):
#   bash tools/pmc_spconv.sh <outfile> [channels] [conv|wgrad] [kernel-name pattern]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; CH=${2:-128}; MODE=${3:-conv}; PAT=${4:-spconv_gather}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_spc
B="python $ROOT/tools/ab/spconv_one.py $CH $MODE"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_spc -o a -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_spc -o b -- $B > /dev/null 2>&1
timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum --kernel-trace -d /tmp/pmc_spc -o c -- $B > /dev/null 2>&1
for f in $(find /tmp/pmc_spc -name '*_results.db' | sort); do echo "== $f"; python $ROOT/tools/pmc_db.py $f "$PAT"; done > $OUT 2>&1
