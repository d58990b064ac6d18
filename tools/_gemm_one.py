import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF
M, N, K = 80000, 256, 256
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / 16; b = torch.zeros(N, device='cuda')
wh, wl, _, _ = UF.split_weight(w)
for _ in range(5): UF.gemm_nt(x, wh, wl, bias=b)
gy = torch.randn(M, N, device='cuda')
for _ in range(5): UF.gemm_wgrad(gy, x)
torch.cuda.synchronize()
