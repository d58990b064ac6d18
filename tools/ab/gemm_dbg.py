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
M, N, K = 98304, 256, 256          # 3 full iterations of 256 blocks x 2 groups
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.zeros(N, device='cuda')
wh, wl, _, _ = UF.split_weight(w)
print(os.environ.get('TAG', ''), f'{timeit(lambda: UF.gemm_nt(x, wh, wl, bias=b)):.1f} us')
