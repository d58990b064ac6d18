mkdir -p gpurun_out/r4t
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_bench_gpu.py tests/test_head_gpu.py tests/test_detector_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4t/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init --no-kernel-timing"
for t in 1 0 1 0 1 0; do UBV_VALUE_CHAIN=$t $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value_chain=$t', round(d['value'],1), round(d['ms_per_step'],3))" >> gpurun_out/r4t/step_ab.txt; done
cat gpurun_out/r4t/tests.txt gpurun_out/r4t/step_ab.txt
