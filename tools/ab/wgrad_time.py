"""f32 weight gradient (dW + db, split-K MFMA kernel + slab sum) at M = 80 000: time per shape (HIP events)."""
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
M = 80000
out = []
for N, K in ((256, 256), (512, 256), (256, 512), (96, 256)):
    gy = torch.randn(M, N, device='cuda'); x = torch.randn(M, K, device='cuda')
    out.append(f'{N}x{K}: {timeit(lambda: UF.gemm_wgrad(gy, x)):.1f}')
print(os.environ.get('TAG', ''), '  '.join(out))
