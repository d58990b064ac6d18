#!/bin/bash
# On the GPU box: everything profiles/r0N_* is made from.  tools/profile_job.sh <name> -> gpurun_out/<name>/
#   gpurun --timeout 3000 -- 'UBV_COMMIT=<sha> bash tools/profile_job.sh r4a'
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py"
# 1. the judged tests
(cd $ROOT && timeout 1500 python -m pytest tests -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt)
# 2. bench lines: default (f32 headline + 16-bit sub-records + gemm / voxel records + CPU baseline), cat-128, one stream
(cd $ROOT && $B --cpu-baseline-plan --extras-file $OUT/bench_extras.json > $OUT/bench.json 2> $OUT/bench.err)
(cd $ROOT && $B --workload LC_cat128 --no-cpu-baseline --no-extras --extras-file $OUT/bench_cat128_extras.json > $OUT/bench_cat128.json 2> $OUT/bench_cat128.err)
for w in C L; do (cd $ROOT && $B --workload $w --no-cpu-baseline --no-extras --extras-file $OUT/bench_${w}_extras.json > $OUT/bench_$w.json 2> $OUT/bench_$w.err); done
(cd $ROOT && $B --single-stream --no-cpu-baseline --no-extras --no-kernel-timing --extras-file '' > $OUT/bench_single_stream.json 2>/dev/null)
# reproducibility of the step's gradients against a one-stream run of the same process, both modes, eager and replayed
(cd $ROOT && UBV_TWO_STREAMS=0 python tools/ab/grad_repro.py 4 2>&1 | grep '^run' > $OUT/grad_repro_one_stream.txt; UBV_TWO_STREAMS=1 python tools/ab/grad_repro.py 12 2>&1 | grep '^run' > $OUT/grad_repro_two_streams.txt; UBV_TWO_STREAMS=1 python tools/ab/grad_repro_graph.py 24 2>&1 | grep '^replay' > $OUT/grad_repro_graph_two_streams.txt; UBV_TWO_STREAMS=0 python tools/ab/grad_repro_graph.py 4 2>&1 | grep '^replay' > $OUT/grad_repro_graph_one_stream.txt; UBV_TWO_STREAMS=1 python tools/ab/fwd_repro_graph.py 24 2>&1 | tail -1 > $OUT/fwd_repro_graph_two_streams.txt; python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent.txt)
# 3. rocprofv3 kernel summary of the default bench command (what roofline.achieved must agree with)
rocprofv3 --kernel-trace -d /tmp/prof_cmd -o cmd -- $B --no-cpu-baseline --extras-file '' > $OUT/bench_profiled.json 2>/dev/null
python $ROOT/tools/db_table.py /tmp/prof_cmd/cmd_results.db 1 80 > $OUT/bench_command_kernel_totals.txt
# 4. per-step kernel tables: eager launches, one stream, 10 + 3 steps
for dt in fp32 bf16; do
  rocprofv3 --kernel-trace -d /tmp/prof_$dt -o e -- $B --dtype $dt --no-graph --single-stream --no-extras \
      --no-cpu-baseline --no-kernel-timing --no-parity --params init --no-ieee-gemm --steps 10 --warmup 3 --extras-file '' > $OUT/bench_${dt}_eager.json 2>/dev/null
  python $ROOT/tools/db_table.py /tmp/prof_$dt/e_results.db 24 60 > $OUT/${dt}_eager_kernel_table.txt
done
# 5. operator micro-benchmarks
for dt in bf16 fp32; do
  python $ROOT/tools/bench_lift.py --dtype $dt > $OUT/bench_lift_$dt.txt 2>&1
  python $ROOT/tools/bench_lift.py --dtype $dt --random-offsets > $OUT/bench_lift_${dt}_spread.txt 2>&1
  python $ROOT/tools/bench_lift.py --dtype $dt --img-hw 800 1440 --dh 16 > $OUT/bench_lift_cat128_$dt.txt 2>&1
done
python $ROOT/tools/bench_gemm.py > $OUT/bench_gemm.txt 2>&1
python $ROOT/tools/bench_gemm_cold.py > $OUT/bench_gemm_cold.txt 2>&1
(cd $ROOT && UBV_GEMM_WS=1 python tools/ab/gemm_ws_ab.py 2>&1 | grep -v '^/opt' > $OUT/gemm_ws_on.txt; UBV_GEMM_WS=0 python tools/ab/gemm_ws_ab.py 2>&1 | grep -v '^/opt' > $OUT/gemm_ws_off.txt)
python $ROOT/tools/bench_backbone.py 2>&1 | grep -v '^/opt' > $OUT/bench_backbone.txt
# 5b. LiDAR front end: middle-encoder kernel table (rebuilt rulebooks), host / device time of the voxel chain,
#     SQ / TCP counters of the 128-channel sparse convolution and its weight gradient
rocprofv3 --kernel-trace -d /tmp/prof_me -o m -- python $ROOT/tools/profile_middle_encoder.py bwd > /dev/null 2>&1
python $ROOT/tools/db_table.py /tmp/prof_me/m_results.db 6 60 > $OUT/middle_encoder_bwd_kernel_table.txt
python $ROOT/tools/ab/vox_host.py 2>&1 | grep -v '^/opt' > $OUT/voxel_chain.txt
for c in 128 64 32 16; do python $ROOT/tools/ab/spconv_one.py $c 2>&1 | grep -v '^/opt'; python $ROOT/tools/ab/spconv_one.py $c wgrad 2>&1 | grep -v '^/opt'; done > $OUT/bench_spconv.txt
bash $ROOT/tools/pmc_spconv.sh $OUT/pmc_spconv128.txt 128 conv spconv_gather
bash $ROOT/tools/pmc_spconv.sh $OUT/pmc_spconv_wgrad128.txt 128 wgrad gemm_wgrad
# 6. HBM traffic per op (PMC passes)
UBV_COMMIT=${UBV_COMMIT:-unrecorded} bash $ROOT/tools/collect_traffic.sh $OUT > $OUT/traffic.log 2>&1
ls -la $OUT
