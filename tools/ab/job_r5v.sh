#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5v
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -x 2>&1 | tail -3
(UBV_GEMM_WS=1 timeout 300 python tools/ab/gemm_ws_ab.py 2>&1 | grep -v '^/opt' > $OUT/gemm_ws_on.txt); cat $OUT/gemm_ws_on.txt
(UBV_GEMM_WS=0 timeout 300 python tools/ab/gemm_ws_ab.py 2>&1 | grep -v "^/opt" > $OUT/gemm_ws_off.txt); cat $OUT/gemm_ws_off.txt
UBV_WS_ABL=16 timeout 300 python - <<'P' 2>&1 | grep -v '^/opt' > $OUT/ws_timing.txt
import os, sys, torch, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from unibev_amd import functional as UF
from unibev_amd._lib import lib
M, N, K, NB = 80000, 256, 256, 8
xs = [torch.randn(M, K, device='cuda') for _ in range(NB)]
ys = [torch.empty(M, N, device='cuda') for _ in range(NB)]
w = torch.randn(N, K, device='cuda') / 16
wh, wl, _, _ = UF.split_weight(w)
for rep in range(3):
    for i in range(NB): UF.gemm_nt(xs[i], wh, wl, out=ys[i])
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 256)()
assert lib().ubv_debug_ws_timing(buf) == 0
import numpy as np
t = np.array(list(buf), dtype=np.int64).reshape(2, 16, 8)
names = ['wait X', 'convert+LDS', 'issue loads', 'barrier', 'MFMA', 'stores']
for b in range(2):
    print('block', (0, 100)[b], ' (cycles of the shader clock; stage start relative to tile 0)')
    for i in range(10):
        s = t[b, i]
        print('  tile %2d  start %7d  ' % (i, s[0] - t[b, 0, 0]) + '  '.join('%s %5d' % (n, s[k + 1] - s[k]) for k, n in enumerate(names)) + '   total %6d' % (s[6] - s[0]))
P
cat $OUT/ws_timing.txt
