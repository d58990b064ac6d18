mkdir -p gpurun_out/r4o
for w in 5 6 7 8 10 14; do
  echo "UBV_CAM32_PER_CU=$w" >> gpurun_out/r4o/cam_sweep.txt
  UBV_CAM32_PER_CU=$w python tools/bench_lift.py --dtype fp32 --only img --iters 30 2>/dev/null | grep "value_camera\|bwd_op" >> gpurun_out/r4o/cam_sweep.txt
done
cat gpurun_out/r4o/cam_sweep.txt | cut -c1-130
