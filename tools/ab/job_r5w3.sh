#!/bin/bash
# PMC counters of the wave-specialised weight gradient (three passes), M = 80 000: 256x256, 96x256, 512x256 averaged
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5w3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_wg
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_wg -o a -- python $ROOT/tools/ab/wgrad_only.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_wg -o b -- python $ROOT/tools/ab/wgrad_only.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_wg -o c -- python $ROOT/tools/ab/wgrad_only.py > /dev/null 2>&1
for f in $(find /tmp/pmc_wg -name '*_results.db' | sort); do echo "== $f"; python $ROOT/tools/pmc_db.py $f gemm_wgrad; done > $O/pmc.txt 2>&1
tail -60 $O/pmc.txt
