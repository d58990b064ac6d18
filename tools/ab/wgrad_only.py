import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
M = 80000
for N, K in ((256, 256), (96, 256), (512, 256)):
    gy = torch.randn(M, N, device='cuda'); x = torch.randn(M, K, device='cuda')
    for _ in range(4): UF.gemm_wgrad(gy, x)
torch.cuda.synchronize()
