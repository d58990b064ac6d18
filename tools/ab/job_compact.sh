mkdir -p gpurun_out/r4r
timeout 900 python -m pytest tests/test_lift_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r4r/tests.txt
python - > gpurun_out/r4r/compact_time.txt 2>&1 <<'P'
import torch, sys
sys.path.insert(0, '.')
from unibev_amd import functional as UF
torch.manual_seed(0)
vis = (torch.rand(6, 40000, device='cuda') < 0.2).to(torch.uint8)
for gw in (0, 200):
    for _ in range(5): UF.compact_visible(vis, gw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): UF.compact_visible(vis, gw)
    b.record(); torch.cuda.synchronize()
    print('grid_w', gw, round(a.elapsed_time(b) / 50 * 1000, 1), 'us per call (launch + kernel)')
P
cat gpurun_out/r4r/tests.txt gpurun_out/r4r/compact_time.txt
