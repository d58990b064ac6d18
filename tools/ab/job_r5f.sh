#!/bin/bash
# (needs the SLP-vectorised build of the library first: make -C unibev_amd/csrc SLP=1 -j8)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5f
mkdir -p $OUT
cd $ROOT
(timeout 600 tools/ubench/pk_mfma_hazard unibev_amd/libunibev_hip.so 200 > $OUT/hazard.txt 2> $OUT/hazard.err; echo "rc $?" >> $OUT/hazard.txt); cat $OUT/hazard.txt; head -20 $OUT/hazard.err
M="alone,beside syn mfma + pk_fma,beside syn mfma + scalar fma,beside syn mfma loop"
(UBV_GEMM_WS=0 UBV_LIB_PATH=$ROOT/unibev_amd/libunibev_hip_slp.so UBV_MODES="$M" timeout 600 python tools/ab/lift_concurrent.py pts 40 2>&1 | grep -v '^/opt' > $OUT/lift_concurrent_slp_victim.txt); cat $OUT/lift_concurrent_slp_victim.txt
for abl in 0 1 2 3 6 7; do echo "UBV_WS_ABL=$abl"; UBV_WS_ABL=$abl timeout 300 python - <<'P' 2>&1 | grep -v '^/opt'
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
P
done > $OUT/ws_ablation.txt; cat $OUT/ws_ablation.txt
