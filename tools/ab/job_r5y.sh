#!/bin/bash
# final check of the round: the judged GPU suite, smoke(), the default bench line (short + extras)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5y
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py --cpu-baseline-plan --extras-file $OUT/bench_extras.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; wc -c $OUT/bench.json; head -c 400 $OUT/bench.json; echo
timeout 600 python bench.py --workload LC_cat128 --no-cpu-baseline --no-extras --extras-file $OUT/bench_cat128_extras.json > $OUT/bench_cat128.json 2>/dev/null; head -c 200 $OUT/bench_cat128.json; echo
