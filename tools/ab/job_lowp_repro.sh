set -x
mkdir -p gpurun_out/r4k
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "adamw" 2>&1 | tail -5 > gpurun_out/r4k/optim_tests.txt
for d in bf16 fp16; do
  UBV_REPRO_DTYPE=$d UBV_TWO_STREAMS=0 timeout 600 python tools/ab/grad_repro.py 6 > gpurun_out/r4k/grad_repro_${d}_one_stream.txt 2>&1
  UBV_REPRO_DTYPE=$d UBV_TWO_STREAMS=1 timeout 600 python tools/ab/grad_repro.py 12 > gpurun_out/r4k/grad_repro_${d}_two_streams.txt 2>&1
done
tail -3 gpurun_out/r4k/*.txt | cut -c1-300
