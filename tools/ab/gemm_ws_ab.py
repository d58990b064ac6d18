#!/usr/bin/env python3
"""f32 Linear GEMMs at M = 80 000, HBM-cold operands (rotating over 8 activation buffers), through ubv_gemm_nt: run once with
UBV_GEMM_WS=0 (tile-per-block kernel, gemm_mfma.hip) and once with UBV_GEMM_WS=1 (weight-stationary persistent kernel,
gemm_ws.hip).  Forms: plain + bias, + residual in place, masked (FFN input gradient), row-periodic term."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF

M, NBUF = 80000, 8


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


print('UBV_GEMM_WS =', os.environ.get('UBV_GEMM_WS', '(default 1)'))
for N, K in ((256, 256), (512, 256), (192, 256), (256, 192), (128, 256), (256, 512)):
    xs = [torch.randn(M, K, device='cuda') for _ in range(NBUF)]
    ys = [torch.empty(M, N, device='cuda') for _ in range(NBUF)]
    rs = [torch.randn(M, N, device='cuda') for _ in range(NBUF)]
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    b = torch.zeros(N, device='cuda')
    pos = torch.randn(40000, N, device='cuda')
    wh, wl, _, _ = UF.split_weight(w)
    hot = timeit(lambda i: UF.gemm_nt(xs[0], wh, wl, bias=b, out=ys[0]))
    cold = timeit(lambda i: UF.gemm_nt(xs[i], wh, wl, bias=b, out=ys[i]))
    res = timeit(lambda i: UF.gemm_nt(xs[i], wh, wl, residual=rs[i], out=rs[i]))
    msk = timeit(lambda i: UF.gemm_nt_act(xs[i], wh, wl, act=2, mask=rs[i], p=0.1))
    rb = timeit(lambda i: UF.gemm_nt(xs[i], wh, wl, bias=b, row_bias=pos, out=ys[i]))
    nb = (M * K + M * N) * 4
    print(f'N={N:4d} K={K:4d}  plain hot {hot:6.1f} us  cold {cold:6.1f} us ({nb / cold / 1e3:5.0f} GB/s)   + residual (in place) '
          f'{res:6.1f} us ({(nb + M * N * 4) / res / 1e3:5.0f} GB/s)   masked {msk:6.1f} us   row-periodic {rb:6.1f} us', flush=True)
    del xs, ys, rs
