"""f32 weight gradient at M = 80 000, 256 x 256 and 512 x 256: run under `rocprofv3 --kernel-trace --stats` to read the
kernel's own duration (the event time of tools/ab/wgrad_time.py includes the slab sum and two launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
M = 80000
NBUF = 6
for N, K in ((256, 256), (512, 256)):
    gys = [torch.randn(M, N, device='cuda') for _ in range(NBUF)]
    xs = [torch.randn(M, K, device='cuda') for _ in range(NBUF)]
    for i in range(4 * NBUF):
        UF.gemm_wgrad(gys[i % NBUF], xs[i % NBUF])
    torch.cuda.synchronize()
