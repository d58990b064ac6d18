mkdir -p gpurun_out/r4y
UBV_TWO_STREAMS=1 timeout 1700 python tools/ab/grad_repro_graph.py 800 2>&1 | grep '^replay' > gpurun_out/r4y/grad_repro_graph_800.txt
python - > gpurun_out/r4y/soak800.txt <<'P'
import re
w=[]
for l in open('gpurun_out/r4y/grad_repro_graph_800.txt'):
    v=[float(x) for x in re.findall(r': ([0-9.e+-]+)[,\n]', l+'\n')]
    if v: w.append(max(v))
import collections
print(len(w), 'two-stream graph replays against the one-stream eager gradients: worst tensor distance', max(w), '; replays above 2e-5:', sum(x > 2e-5 for x in w))
P
cat gpurun_out/r4y/soak800.txt
