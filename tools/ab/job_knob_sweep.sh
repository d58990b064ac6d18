# the step's samples/s under a few launch-shape knobs (f32, init parameters, 30 steps each)
mkdir -p gpurun_out/r4m
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init --no-kernel-timing"
run() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))" >> gpurun_out/r4m/sweep.txt; }
run base A=1
run base2 A=1
run wgrad256 UBV_WGRAD_BLOCKS=256
run wgrad384 UBV_WGRAD_BLOCKS=384
run wgrad768 UBV_WGRAD_BLOCKS=768
run wgrad1024 UBV_WGRAD_BLOCKS=1024
run gridw2 UBV_GRID_WAVES=2
run gridw4 UBV_GRID_WAVES=4
run gemm_nt256 UBV_GEMM_NT=256
run cam32w6 UBV_CAM32_PER_CU=6
run cam32w8 UBV_CAM32_PER_CU=8
run base3 A=1
cat gpurun_out/r4m/sweep.txt
