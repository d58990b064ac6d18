#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5a3; mkdir -p $O
PREV=$PWD/unibev_amd/libunibev_hip_prev.so
timeout 1500 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for lib in prev new prev new; do
  if [ $lib = prev ]; then export UBV_LIB_PATH=$PREV; else unset UBV_LIB_PATH; fi
  for keep in 1 0; do
    UBV_KEEP_RULEBOOKS=$keep TAG="$lib keep=$keep" timeout 300 python tools/ab/middle_encoder_time.py 2>&1 | tail -1 | tee -a $O/ab.txt
  done
done
unset UBV_LIB_PATH
for c in 128 64 32 16; do python tools/ab/spconv_one.py $c 2>&1 | grep -v '^/opt'; done | tee -a $O/ab.txt
