import sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
device = torch.device('cuda', 0)
torch.manual_seed(0); np.random.seed(0)
head, _ = bench.build_head('LC_cnw', device); head.train()
img, pts, metas = bench.synth_inputs('LC_cnw', 2, torch.bfloat16, device, 0)
cot = torch.randn(200 * 200, 2, 256, device=device) / 200.0
def fwd():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        return head.forward_bev(img, pts, metas)
for _ in range(3):
    (fwd().float() * cot).sum().backward()
torch.cuda.synchronize()
N = 10
hf = wf = hb = wb = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fwd(); hf += time.perf_counter() - t0
    torch.cuda.synchronize(); wf += time.perf_counter() - t0
    t0 = time.perf_counter(); (out.float() * cot).sum().backward(); hb += time.perf_counter() - t0
    torch.cuda.synchronize(); wb += time.perf_counter() - t0
print('forward : host %.2f ms, wall %.2f ms' % (1e3 * hf / N, 1e3 * wf / N))
print('backward: host %.2f ms, wall %.2f ms' % (1e3 * hb / N, 1e3 * wb / N))
