ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_fp32 -o e -- python $ROOT/bench.py --dtype fp32 --no-graph --single-stream --no-extras --no-cpu-baseline --no-kernel-timing --no-parity --params init --no-ieee-gemm --steps 10 --warmup 3 > $OUT/bench_fp32_eager.json 2>/dev/null
python $ROOT/tools/db_table.py /tmp/prof_fp32/e_results.db 24 70 > $OUT/fp32_eager_kernel_table.txt
head -3 $OUT/fp32_eager_kernel_table.txt
