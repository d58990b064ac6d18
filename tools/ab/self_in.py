import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
M, C = 80000, 256
x = torch.randn(M, C, device='cuda'); w = torch.randn(352, C, device='cuda') / 16; b = torch.zeros(352, device='cuda')
rb = torch.randn(40000, 96, device='cuda')
wh, wl, wth, wtl = UF.split_weight(w)
whv, wlv, wtv_h, wtv_l = UF.split_weight(w[:256].contiguous()); who, wlo, wto_h, wto_l = UF.split_weight(w[256:].contiguous())
gv = torch.randn(M, 256, device='cuda'); gol = torch.randn(M, 96, device='cuda'); ga = torch.randn(M, C, device='cuda')
print('fwd fused   ', timeit(lambda: UF.gemm_nt_dual(x, wh, wl, bias=b, y2_cols=96, row_bias=rb)))
print('fwd separate', timeit(lambda: (UF.gemm_nt(x, whv, wlv, bias=b[:256]), UF.gemm_nt(x, who, wlo, bias=b[256:], row_bias=rb))))
print('dgrad fused   ', timeit(lambda: UF.gemm_nt_dual(gv, wth, wtl, x2=gol, residual=ga)))
def sep():
    t = UF.gemm_nt(gol, wto_h, wto_l, residual=ga)
    return UF.gemm_nt(gv, wtv_h, wtv_l, residual=t, out=t)
print('dgrad separate', timeit(sep))
print('wgrad fused   ', timeit(lambda: UF.gemm_wgrad_dual(gv, gol, x)))
print('wgrad separate', timeit(lambda: (UF.gemm_wgrad(gv, x), UF.gemm_wgrad(gol, x))))
