mkdir -p gpurun_out/r4z
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_bench_gpu.py tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4z/tests_grb.txt
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))" >> gpurun_out/r4z/tests_grb.txt
cat gpurun_out/r4z/tests_grb.txt
