#!/bin/bash
# per-step kernel table of cfg5 (L+C cat-128, 1440x800 images, C = 128), f32, eager, one stream
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5c5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_c5 -o e -- python $ROOT/bench.py --workload LC_cat128 --dtype fp32 --no-graph --single-stream --no-extras --no-cpu-baseline --no-kernel-timing --no-parity --params init --no-ieee-gemm --steps 10 --warmup 3 --extras-file '' > $O/bench_eager.json 2>/dev/null
python $ROOT/tools/db_table.py /tmp/prof_c5/e_results.db 24 60 > $O/cat128_fp32_eager_kernel_table.txt
head -45 $O/cat128_fp32_eager_kernel_table.txt | cut -c1-170
