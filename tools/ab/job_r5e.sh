#!/bin/bash
# (needs the SLP-vectorised build of the library first: make -C unibev_amd/csrc SLP=1 -j8)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5e
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -x 2>&1 | tail -5 > $OUT/gemm_tests.txt; cat $OUT/gemm_tests.txt
(UBV_GEMM_WS=0 timeout 300 python tools/ab/gemm_ws_ab.py 2>&1 | grep -v '^/opt' > $OUT/gemm_ws_off.txt); cat $OUT/gemm_ws_off.txt
(UBV_GEMM_WS=1 timeout 300 python tools/ab/gemm_ws_ab.py 2>&1 | grep -v '^/opt' > $OUT/gemm_ws_on.txt); cat $OUT/gemm_ws_on.txt
M="alone,beside gemm_nt,beside gemm_nt (other build),beside syn pk_fma loop,beside syn mfma + pk_fma"
(UBV_GEMM_WS=0 UBV_LIB_PATH=$ROOT/unibev_amd/libunibev_hip_slp.so UBV_OTHER_LIB=$ROOT/unibev_amd/libunibev_hip.so UBV_MODES="$M" timeout 600 python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent_slp_victim.txt)
(UBV_GEMM_WS=0 UBV_OTHER_LIB=$ROOT/unibev_amd/libunibev_hip_slp.so UBV_MODES="$M" timeout 600 python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent_noslp_victim.txt)
cat $OUT/lift_concurrent_slp_victim.txt $OUT/lift_concurrent_noslp_victim.txt
