#!/usr/bin/env python3
"""Projection GEMMs at M = 80 000: the hand-written MFMA kernel (ubv_gemm_nt) next to the library path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


M = 80000
for dt in (torch.float32, torch.bfloat16):
    for N, K in ((256, 256), (192, 256), (96, 256), (512, 256), (256, 512)):
        x = torch.randn(M, K, device='cuda').to(dt)
        w = (torch.randn(N, K, device='cuda') / K ** 0.5)
        b = torch.zeros(N, device='cuda')
        if dt == torch.float32:
            wh, wl, _, _ = UF.split_weight(w)
            mine = timeit(lambda: UF.gemm_nt(x, wh, wl, bias=b))
            lib = timeit(lambda: UF.linear_forward(x, w, b))
        else:
            w16 = w.to(dt)
            mine = timeit(lambda: UF.gemm_nt(x, w16, bias=b))
            lib = timeit(lambda: UF.linear_forward(x, w16, b.to(dt)))
        nbytes = (M * K + M * N) * x.element_size()
        print(f'{str(dt):16s} N={N:4d} K={K:4d}  mfma kernel {mine:7.1f} us ({nbytes / mine / 1e3:6.0f} GB/s)   '
              f'library {lib:7.1f} us ({nbytes / lib / 1e3:6.0f} GB/s)')

print('--- weight gradient (dW + db): MFMA split-K kernel + slab sum vs strided-batched library GEMM + reduce')
from unibev_amd.linear import _splits
for dt in (torch.float32, torch.bfloat16):
    for N, K in ((256, 256), (192, 256), (96, 256), (512, 256), (256, 512)):
        gy = torch.randn(M, N, device='cuda').to(dt)
        x = torch.randn(M, K, device='cuda').to(dt)
        mine = timeit(lambda: UF.gemm_wgrad(gy, x))
        s = _splits(M)

        def lib():
            part = torch.bmm(gy.view(s, M // s, -1).transpose(1, 2), x.view(s, M // s, -1))
            return UF.linear_grad_reduce(gy, part)
        libt = timeit(lib)
        nbytes = (M * K + M * N) * x.element_size()
        print(f'{str(dt):16s} N={N:4d} K={K:4d}  mfma wgrad {mine:7.1f} us ({nbytes / mine / 1e3:6.0f} GB/s)   '
              f'library {libt:7.1f} us')
