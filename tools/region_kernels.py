#!/usr/bin/env python3
"""Framework (aten) operators that run inside the two-stream fork/join windows of one f32 training step
(unibev_amd.debug.ForeignKernelLog): what tests/test_region_gpu.py asserts to be empty, with call sites."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                                   # noqa: E402
from unibev_amd.debug import ForeignKernelLog                       # noqa: E402
from unibev_amd.modules import transformer as TR                    # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
head, _ = B.build_head('LC_cnw', dev)
img, pts, metas = B.synth_inputs('LC_cnw', 2, torch.float32, dev, 0)
params = [p for p in head.parameters() if p.requires_grad]
cot = torch.randn(200 * 200, 2, 256, device=dev) / 200.0
head.transformer.forced_flags = (1, 1)


def step(log=None):
    for p in params + (img or []) + (pts or []):
        p.grad = None
    out = head.forward_bev(img, pts, metas)
    if log is not None:
        log.mark('backward')
    out.backward(cot)


for _ in range(2):
    step()
torch.cuda.synchronize()
with ForeignKernelLog(TR._side_stream(dev)) as log:
    step(log)
torch.cuda.synchronize()
print('windows', log.summary())
bad = log.offenders()
for k, v in sorted(bad.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(v, k)
print('offending launches per step:', sum(bad.values()))
