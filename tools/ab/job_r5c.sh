#!/bin/bash
# (needs the SLP-vectorised build of the library first: make -C unibev_amd/csrc SLP=1 -j8)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5c
mkdir -p $OUT
cd $ROOT
(UBV_LIB_PATH=$ROOT/unibev_amd/libunibev_hip_slp.so UBV_MODES="alone,beside gemm_nt,beside syn cvt_pk_bf16 loop,beside syn mfma 8 accumulators,beside syn copy,beside syn mfma loop" timeout 600 python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent_slp2.txt)
cat $OUT/lift_concurrent_slp2.txt
timeout 900 python -m pytest tests/test_bench_gpu.py -q -m gpu -x 2>&1 | tail -3
