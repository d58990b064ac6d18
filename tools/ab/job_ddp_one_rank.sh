# the multi-GPU code path on one GPU: process group + flat-gradient all-reduce over RCCL with one rank
set -x
mkdir -p gpurun_out/r4l
export HSA_ENABLE_IPC_MODE_LEGACY=0 UBV_FORCE_DDP=1
for ex in auto split; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 1 --steps 10 --warmup 3 --exchange $ex --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init > gpurun_out/r4l/bench_ddp1_$ex.json 2> gpurun_out/r4l/bench_ddp1_$ex.err
  echo rc=$?
  tail -c 600 gpurun_out/r4l/bench_ddp1_$ex.err
done
