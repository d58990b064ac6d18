#!/usr/bin/env python3
"""1000 launches of the GRID-plan backward at the BASELINE size (SCA-pts instance, bs=2, bf16) with the
grad_value chain on a side stream (UBV_LIFT_TWO_STREAM=1) against the single-stream result: counts
launches whose query-side gradients differ bitwise."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_lift import instance
from unibev_amd import functional as UF

value, offlog, ref, vis0, count, gout, geom, is_grid, center = instance('pts', 2, torch.bfloat16, 'cuda')
B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh = geom
offlog = offlog.to(torch.bfloat16).requires_grad_()
value.requires_grad_()


def run():
    value.grad = offlog.grad = None
    UF.bev_lift(value, offlog, ref, Nc, (fh, fw), H, P, query_grid=(qh, qw), ref_is_grid=True).backward(gout)
    return offlog.grad.clone(), value.grad.clone()


os.environ['UBV_LIFT_TWO_STREAM'] = '0'
base = run()
for mode in ('0', '1'):
    os.environ['UBV_LIFT_TWO_STREAM'] = mode
    bad_q = bad_v = 0
    worst = 0.0
    for _ in range(1000):
        g, gv = run()
        bad_q += int(not torch.equal(g, base[0]))
        if not torch.equal(gv, base[1]):
            bad_v += 1
            worst = max(worst, float((gv.float() - base[1].float()).abs().max()))
    print(f'two_stream={mode}: launches with different d(offsets|logits): {bad_q}/1000; '
          f'different grad_value: {bad_v}/1000 (max abs diff {worst:.3g}, bucket arrival order)')
