#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5w
mkdir -p $OUT
cd $ROOT
env -u RANK -u WORLD_SIZE -u LOCAL_RANK timeout 1200 python bench.py --gpus 1 --launcher spawn --exchange split --steps 5 --warmup 2 --extras-file $OUT/extras.json > $OUT/bench_split_spawn.json 2> $OUT/bench_split_spawn.err; echo "rc $?"; tail -c 600 $OUT/bench_split_spawn.err | grep -v "^#extras" | tail -5
python - <<P
import json
d=json.loads(open('$OUT/bench_split_spawn.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['gradient_exchange'], d['config']['rccl_ranks'], d['phases'], d.get('lowp'), d.get('spread_value'), d.get('k1_operator'))
P
