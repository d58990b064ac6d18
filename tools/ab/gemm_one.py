import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
M = int(os.environ.get('M', 80000)); N = int(os.environ.get('N', 256)); K = int(os.environ.get('K', 256))
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.zeros(N, device='cuda')
wh, wl, _, _ = UF.split_weight(w)
for _ in range(int(os.environ.get('REPS', 12))):
    y = UF.gemm_nt(x, wh, wl, bias=b)
torch.cuda.synchronize()
