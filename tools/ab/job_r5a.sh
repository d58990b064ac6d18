#!/bin/bash
# round 5, first job: hazard reproducer, GPU tests, default bench line (short-line check)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5a
mkdir -p $OUT
cd $ROOT
(timeout 600 tools/ubench/pk_mfma_hazard unibev_amd/libunibev_hip.so 400 > $OUT/hazard.txt 2> $OUT/hazard.err; echo "rc $?" >> $OUT/hazard.txt)
(GPU_MAX_HW_QUEUES=1 timeout 600 tools/ubench/pk_mfma_hazard unibev_amd/libunibev_hip.so 200 > $OUT/hazard_q1.txt 2> $OUT/hazard_q1.err; echo "rc $?" >> $OUT/hazard_q1.txt)
cat $OUT/hazard.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt
timeout 900 python bench.py --extras-file $OUT/bench_extras.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench.json; head -c 1500 $OUT/bench.json
