#!/bin/bash
# (needs the SLP-vectorised build of the library first: make -C unibev_amd/csrc SLP=1 -j8)
# round 5: hazard study, stage 2 — the REAL victim (lifting kernel built with the SLP vectoriser: packed f32) beside synthetic
# co-runners, and the default (no packed f32) build as control; then the GPU tests
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5b
mkdir -p $OUT
cd $ROOT
(UBV_LIB_PATH=$ROOT/unibev_amd/libunibev_hip_slp.so timeout 600 python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent_slp.txt)
(timeout 600 python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent_noslp.txt)
cat $OUT/lift_concurrent_slp.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/tests.txt 2>&1; tail -8 $OUT/tests.txt
