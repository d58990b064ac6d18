mkdir -p gpurun_out/r4o
for t in 0 1; do
  UBV_CAM_TILED=$t python tools/bench_lift.py --dtype fp32 --only img --iters 30 2>/dev/null | grep -v amdgpu > gpurun_out/r4o/lift_img_fp32_tiled$t.txt
  UBV_CAM_TILED=$t python tools/bench_lift.py --dtype bf16 --only img --iters 30 2>/dev/null | grep -v amdgpu > gpurun_out/r4o/lift_img_bf16_tiled$t.txt
done
timeout 900 python -m pytest tests/test_lift_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4o/tests.txt
tail -n 6 gpurun_out/r4o/*.txt
