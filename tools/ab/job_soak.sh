mkdir -p gpurun_out/r4y
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extras --no-parity --no-ieee-gemm --dtype fp32 --params init --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('400 steps:', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms per step')" > gpurun_out/r4y/soak.txt
UBV_TWO_STREAMS=1 timeout 900 python tools/ab/grad_repro_graph.py 80 2>&1 | grep '^replay' > gpurun_out/r4y/grad_repro_graph_80.txt
python - >> gpurun_out/r4y/soak.txt <<'P'
import re
w=[]
for l in open('gpurun_out/r4y/grad_repro_graph_80.txt'):
    v=[float(x) for x in re.findall(r': ([0-9.e+-]+)[,\n]', l)]
    if v: w.append(max(v))
print(len(w), 'graph replays (forward + backward, two streams) against the one-stream eager gradients: worst tensor distance', max(w))
P
cat gpurun_out/r4y/soak.txt
