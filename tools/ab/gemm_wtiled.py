import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
def timeit(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
M = 80000
tiled = os.environ.get('UBV_GEMM_WTILED', '0') == '1'
out = []
for N, K in ((256, 256), (96, 256), (512, 256), (256, 512)):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.zeros(N, device='cuda')
    wh, wl, _, _ = UF.split_weight(w)
    ref = UF.gemm_nt(x, wh, wl, bias=b) if not tiled else None
    if tiled:
        os.environ['UBV_GEMM_WTILED'] = '1'
        wh2 = wh.view(N, K // 32, 32).permute(1, 0, 2).contiguous().view(N, K)
        wl2 = wl.view(N, K // 32, 32).permute(1, 0, 2).contiguous().view(N, K)
        y = UF.gemm_nt(x, wh2, wl2, bias=b)
        err = (y - x @ w.t()).abs().max().item()
        out.append(f'{N}<-{K}: {timeit(lambda: UF.gemm_nt(x, wh2, wl2, bias=b)):.1f} (err {err:.1e})')
    else:
        err = (ref - x @ w.t()).abs().max().item()
        out.append(f'{N}<-{K}: {timeit(lambda: UF.gemm_nt(x, wh, wl, bias=b)):.1f} (err {err:.1e})')
print(os.environ.get('TAG', ''), '  '.join(out))
