#!/usr/bin/env python3
"""cProfile of the host side of bench.py's training step (where the Python time goes once the
step is host-bound).   python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    device = torch.device('cuda', 0)
    torch.manual_seed(0)
    np.random.seed(0)
    head, _ = bench.build_head('LC_cnw', device)
    head.train()
    img, pts, metas = bench.synth_inputs('LC_cnw', 2, torch.bfloat16, device, 0)
    params = [p for p in head.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.01, fused=True)
    cot = torch.randn(200 * 200, 2, 256, device=device) / 200.0

    def step():
        opt.zero_grad(set_to_none=True)
        for x in img + pts:
            x.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            fused = head.forward_bev(img, pts, metas)
        (fused.float() * cot).sum().backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    st = pstats.Stats(pr, stream=out)
    st.sort_stats('tottime').print_stats(45)
    print(out.getvalue())


if __name__ == '__main__':
    main()
