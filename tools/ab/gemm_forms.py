"""f32 gemm_nt at M = 80 000 in the forms the step uses: plain, + residual, relu/dropout epilogue, masked input gradient."""
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
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda')
    wh, wl, _, _ = UF.split_weight(w)
    ref = x @ w.t() + b
    y = UF.gemm_nt(x, wh, wl, bias=b); e0 = (y - ref).abs().max().item()
    y = UF.gemm_nt(x, wh, wl, bias=b, residual=r); e1 = (y - ref - r).abs().max().item()
    mask = (torch.rand(M, N, device='cuda') > 0.3).float()
    y = UF.gemm_nt_act(x, wh, wl, None, act=2, mask=mask, p=0.1); e2 = (y - (x @ w.t()) * mask / 0.9).abs().max().item()
    y = UF.gemm_nt_act(x, wh, wl, b, act=1, p=0.0); e3 = (y - ref.relu()).abs().max().item()
    t = [timeit(lambda: UF.gemm_nt(x, wh, wl, bias=b)), timeit(lambda: UF.gemm_nt(x, wh, wl, bias=b, residual=r)),
         timeit(lambda: UF.gemm_nt_act(x, wh, wl, None, act=2, mask=mask, p=0.1)),
         timeit(lambda: UF.gemm_nt_act(x, wh, wl, b, act=1, p=0.1, seed=5))]
    out.append(f'{N}<-{K}: plain {t[0]:.1f} +res {t[1]:.1f} masked {t[2]:.1f} relu-drop {t[3]:.1f} (err {e0:.0e} {e1:.0e} {e2:.0e} {e3:.0e})')
print(os.environ.get('TAG', ''), '\n  '.join(out))
