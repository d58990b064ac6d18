#!/usr/bin/env python3
"""How much do an input-gradient GEMM and a weight-gradient GEMM gain from running side by side?
Each loops on its own stream (no events between them); compared with the two loops back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF

M, N, K = 80000, 256, 256
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(M, K, device='cuda').to(dt)
    gy = torch.randn(M, N, device='cuda').to(dt)
    w = torch.randn(N, K, device='cuda') / 16
    if dt == torch.float32:
        wh, wl, _, _ = UF.split_weight(w)
        dgrad = lambda: UF.gemm_nt(gy, wh, wl)
    else:
        w16 = w.to(dt)
        dgrad = lambda: UF.gemm_nt(gy, w16)
    wgrad = lambda: UF.gemm_wgrad(gy, x)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    n = 40

    def run(fa, fb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fa:
            with torch.cuda.stream(sa):
                for _ in range(n):
                    fa()
        if fb:
            with torch.cuda.stream(sb):
                for _ in range(n):
                    fb()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / n
    for _ in range(2):
        run(dgrad, wgrad)
    a, b, both = run(dgrad, None), run(None, wgrad), run(dgrad, wgrad)
    print(f'{dt}: dgrad alone {a:.1f} us, wgrad alone {b:.1f} us, side by side {both:.1f} us per pair '
          f'(sum {a + b:.1f}: {100 * (1 - both / (a + b)):.0f} % saved)')
