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
N = int(os.environ.get('N', 256)); K = int(os.environ.get('K', 256))
w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.zeros(N, device='cuda')
wh, wl, _, _ = UF.split_weight(w)
out = []
for blocks_per_cu in (0.5, 1, 2, 3, 4, 4.9, 6, 9, 12):
    M = int(blocks_per_cu * 256 * 128 / (N // 128))
    x = torch.randn(M, K, device='cuda')
    t = timeit(lambda: UF.gemm_nt(x, wh, wl, bias=b))
    out.append(f'{blocks_per_cu}/CU M={M}: {t:.1f} us ({(M * (K + N) * 4) / t / 1e6:.2f} TB/s)')
print(os.environ.get('TAG', ''), '\n'.join(out))
