mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_lift_gpu.py -x -q -m gpu -k "visible_lists" 2>&1 | tail -5 > gpurun_out/r4o/test_lists.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init --no-kernel-timing"
for t in 1 0 1 0; do UBV_CAM_TILED=$t $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled=$t', round(d['value'],1), round(d['ms_per_step'],3))" >> gpurun_out/r4o/step_ab.txt; done
cat gpurun_out/r4o/test_lists.txt gpurun_out/r4o/step_ab.txt
