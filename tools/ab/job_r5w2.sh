#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5w2; mkdir -p $O
for abl in 0 1 2 3 6 7; do
  UBV_WGRAD_PW=4 UBV_WGRAD_ABL=$abl TAG="abl=$abl" timeout 120 python tools/ab/wgrad_time.py 2>&1 | tail -1 | tee -a $O/abl3.txt
done
