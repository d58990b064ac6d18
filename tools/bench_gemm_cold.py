#!/usr/bin/env python3
"""f32 projection GEMM / weight gradient at M = 80 000 with HBM-COLD operands: the calls rotate over enough distinct
activation buffers (> 256 MiB Infinity Cache) that every call reads its X from HBM, as in the training step (the
plain micro-benchmark re-reads one 82 MB X out of the Infinity Cache).  A device copy of the same bytes rides along."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF

M = 80000
NBUF = int(os.environ.get('NBUF', '8'))


def timeit(fn, n=NBUF * 3):
    for i in range(NBUF):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % NBUF)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for N, K in ((256, 256), (512, 256), (256, 512), (96, 256)):
    xs = [torch.randn(M, K, device='cuda') for _ in range(NBUF)]
    ys = [torch.empty(M, N, device='cuda') for _ in range(NBUF)]
    gys = [torch.randn(M, N, device='cuda') for _ in range(NBUF)]
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    b = torch.zeros(N, device='cuda')
    wh, wl, _, _ = UF.split_weight(w)
    hot = timeit(lambda i: UF.gemm_nt(xs[0], wh, wl, bias=b))
    cold = timeit(lambda i: UF.gemm_nt(xs[i], wh, wl, bias=b))
    wg_hot = timeit(lambda i: UF.gemm_wgrad(gys[0], xs[0]))
    wg_cold = timeit(lambda i: UF.gemm_wgrad(gys[i], xs[i]))
    cp = timeit(lambda i: ys[i][:, :min(N, K)].copy_(xs[i][:, :min(N, K)])) if N == K else float('nan')
    nb = (M * K + M * N) * 4
    print(f'N={N:4d} K={K:4d}  gemm_nt hot {hot:6.1f} us  cold {cold:6.1f} us ({nb / cold / 1e3:5.0f} GB/s)   '
          f'wgrad hot {wg_hot:6.1f}  cold {wg_cold:6.1f} us ({nb / wg_cold / 1e3:5.0f} GB/s)   copy {cp:6.1f} us')
