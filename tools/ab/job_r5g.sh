#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5g
mkdir -p $OUT
cd $ROOT
(timeout 600 tools/ubench/pk_mfma_hazard - 200 > $OUT/hazard.txt 2> $OUT/hazard.err; echo "rc $?" >> $OUT/hazard.txt); cut -c1-28,29-1000 $OUT/hazard.txt | awk -F'|' '{print $1 "|" $6 "|" $7}'; head -8 $OUT/hazard.err
for abl in 0 8 9 1; do echo "UBV_WS_ABL=$abl"; UBV_WS_ABL=$abl timeout 300 python - <<'P' 2>&1 | grep -v '^/opt'
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from unibev_amd import functional as UF
M, N, K, NB = 80000, 256, 256, 8
xs = [torch.randn(M, K, device='cuda') for _ in range(NB)]
ys = [torch.empty(M, N, device='cuda') for _ in range(NB)]
w = torch.randn(N, K, device='cuda') / 16
wh, wl, _, _ = UF.split_weight(w)
def t(fn, n=24):
    for i in range(NB): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % NB)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
print('  hot %.1f us   cold %.1f us' % (t(lambda i: UF.gemm_nt(xs[0], wh, wl, out=ys[0])), t(lambda i: UF.gemm_nt(xs[i], wh, wl, out=ys[i]))))
ref = xs[1].double() @ w.double().t()
print('  max err', float((UF.gemm_nt(xs[1], wh, wl).double() - ref).abs().max() / ref.abs().max()))
P
done > $OUT/ws_ablation.txt; cat $OUT/ws_ablation.txt
