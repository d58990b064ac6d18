#!/bin/bash
# One parametrised GPU job instead of a script per experiment (rounds 4 - 5 left 61 of those; they are in the git history
# up to d67aac5).  Runs ON the GPU box; every part writes under gpurun_out/<name>/ and prints a short tail.
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_job.sh r6a tests bench'
#   gpurun --timeout 900  -- 'bash tools/gpu_job.sh r6b knobs UBV_TILE_CENTER=0 UBV_TILE_CENTER=1'
#   gpurun --timeout 900  -- 'bash tools/gpu_job.sh r6c lift fp32 -- pmc tools/bench_lift.py lift_tile'
#
# parts (in the order given; "--" separates a part that takes free arguments from the next one):
#   tests [pytest args]      pytest -m gpu (default: the whole suite)                      -> tests.txt
#   bench [bench.py args]    the default bench line + extras                                -> bench.json, bench_extras.json
#   short [bench.py args]    f32 / init parameters / 30 steps, nothing else (A/B of a step) -> short.txt
#   knobs ENV=V ...          `short` once per environment setting, one line each           -> knobs.txt
#   lift [dtype] [args]      tools/bench_lift.py (default fp32), also --random-offsets     -> lift_<dtype>.txt
#   table [dtype]            rocprofv3 kernel table of the eager one-stream step            -> <dtype>_eager_kernel_table.txt
#   pmc SCRIPT FILTER        three SQ counter passes over `python SCRIPT`, kernels matching FILTER -> pmc.txt
#   traffic                  tools/collect_traffic.sh (HBM bytes per sampling op)           -> traffic.*
#   py SCRIPT [args]         python SCRIPT args                                            -> py_<script>.txt
#   profile                  tools/profile_job.sh <name> (everything profiles/rNN_* is made from)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=${1:?job name}; shift
OUT=$ROOT/gpurun_out/$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHORT="python $ROOT/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init --no-kernel-timing --extras-file ''"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"; }
args=()
take() { args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do case "$1" in tests|bench|short|knobs|lift|table|pmc|traffic|py|profile) break;; esac; args+=("$1"); shift; done; rest=("$@"); [ "${rest[0]}" == "--" ] && rest=("${rest[@]:1}"); }
while [ $# -gt 0 ]; do
  part=$1; shift; take "$@"; set -- "${rest[@]}"
  case $part in
    tests) (cd $ROOT && timeout 2400 python -m pytest ${args[@]:-tests} -q -m gpu -x > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt);;
    bench) (cd $ROOT && python bench.py --extras-file $OUT/bench_extras.json "${args[@]}" > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json);;
    short) (cd $ROOT && $SHORT "${args[@]}" 2>/dev/null | line short | tee -a $OUT/short.txt);;
    knobs) for kv in "${args[@]}"; do (cd $ROOT && env $kv $SHORT $UBV_SHORT_ARGS 2>/dev/null | line "$kv" | tee -a $OUT/knobs.txt); done;;
    lift) dt=${args[0]:-fp32}; extra=("${args[@]:1}")
          python $ROOT/tools/bench_lift.py --dtype $dt "${extra[@]}" > $OUT/lift_$dt.txt 2>&1
          python $ROOT/tools/bench_lift.py --dtype $dt --random-offsets "${extra[@]}" > $OUT/lift_${dt}_spread.txt 2>&1
          grep -h 'B=' $OUT/lift_$dt.txt $OUT/lift_${dt}_spread.txt;;
    table) dt=${args[0]:-fp32}; rm -rf /tmp/prof_$dt
          rocprofv3 --kernel-trace -d /tmp/prof_$dt -o e -- python $ROOT/bench.py --dtype $dt --no-graph --single-stream --no-extras \
              --no-cpu-baseline --no-kernel-timing --no-parity --params init --no-ieee-gemm --steps 10 --warmup 3 --extras-file '' > /dev/null 2>&1
          python $ROOT/tools/db_table.py /tmp/prof_$dt/e_results.db 24 60 > $OUT/${dt}_eager_kernel_table.txt; head -30 $OUT/${dt}_eager_kernel_table.txt | cut -c1-150;;
    pmc) script=${args[0]}; filt=${args[1]}; rm -rf /tmp/pmc_job
          timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_job -o a -- python $ROOT/$script "${args[@]:2}" > /dev/null 2>&1
          timeout 300 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_job -o b -- python $ROOT/$script "${args[@]:2}" > /dev/null 2>&1
          timeout 300 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_job -o c -- python $ROOT/$script "${args[@]:2}" > /dev/null 2>&1
          for f in $(find /tmp/pmc_job -name '*_results.db' | sort); do echo "== $f"; python $ROOT/tools/pmc_db.py $f $filt; done > $OUT/pmc.txt 2>&1; tail -40 $OUT/pmc.txt;;
    traffic) UBV_COMMIT=${UBV_COMMIT:-unrecorded} bash $ROOT/tools/collect_traffic.sh $OUT > $OUT/traffic.log 2>&1; tail -5 $OUT/traffic.log;;
    py) s=${args[0]}; (cd $ROOT && timeout 1200 python $s "${args[@]:1}" 2>&1 | grep -v '^/opt' | tee $OUT/py_$(basename $s .py).txt | tail -40);;
    profile) bash $ROOT/tools/profile_job.sh $NAME;;
    *) echo "unknown part $part"; exit 2;;
  esac
done
