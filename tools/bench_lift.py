#!/usr/bin/env python3
"""Micro-benchmark of the BEV lifting kernels at the BASELINE instance shapes (bs=2 by default).

Reports per-launch time (HIP events around the C-ABI call, median of N) and algorithmic GB/s
(compulsory bytes) for forward and backward of
  self-attn (200x200 map, P=4), SCA-pts (180x180 map, P=8), SCA-img (6 x 8x22 maps, P=8).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF      # noqa: E402
from unibev_amd import synthetic as syn      # noqa: E402


def instance(name, B, dtype, dev, init_like=True, img_hw=(256, 704), Dh=32):
    H, qh, qw = 8, 200, 200
    Nq, C = qh * qw, H * Dh
    g = torch.Generator(device='cpu').manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(qh), torch.arange(qw), indexing='ij')
    grid = torch.stack(((xs + 0.5) / qw, (ys + 0.5) / qh), -1).view(1, 1, Nq, 1, 2).float()
    if name == 'self':
        Nc, fh, fw, P, Z = 1, 200, 200, 4, 1
        ref = grid.expand(1, B, Nq, Z, 2).contiguous()
        vis0 = count = None
    elif name == 'pts':
        Nc, fh, fw, P, Z = 1, 180, 180, 8, 4
        ref = grid.expand(1, B, Nq, Z, 2).contiguous()
        vis0 = count = None
    else:
        Nc, P, Z = 6, 8, 4
        fh, fw = img_hw[0] // 32, img_hw[1] // 32
        from unibev_amd.modules.encoders import pillar_axes
        l2i = torch.from_numpy(np.stack([syn.camera_rig(6, img_hw)] * B)).float().to(dev)
        axes = pillar_axes(qh, qw, 8, 4, dev)
        ref, mask, vis0, count = UF.point_sampling(l2i, *axes, syn.PC_RANGE, img_hw)
    # offsets as init_weights leaves them (compass directions x point index) + small noise
    th = torch.arange(H).float() * (2 * np.pi / H)
    d = torch.stack((th.cos(), th.sin()), -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    off = d.view(1, 1, H, 1, 2) * torch.arange(1, P + 1).view(1, 1, 1, P, 1).float()
    off = off.expand(B, Nq, H, P, 2) + 0.3 * torch.randn(B, Nq, H, P, 2, generator=g)
    if not init_like:
        off = 3.0 * torch.randn(B, Nq, H, P, 2, generator=g)
    logits = torch.randn(B, Nq, H * P, generator=g)
    offlog = torch.cat((off.reshape(B, Nq, -1), logits), -1).to(dev)
    value = torch.randn(B * Nc, fh * fw, C, generator=g).to(dev, dtype)
    gout = torch.randn(B, Nq, C, generator=g).to(dev, dtype)
    geom = (B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh)
    center = (d.view(H, 1, 2) * torch.arange(1, P + 1).view(1, P, 1).float()).reshape(-1).to(dev)
    return value, offlog, ref.to(dev), vis0, count, gout, geom, (name != 'img'), center


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=2)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--random-offsets', action='store_true')
    ap.add_argument('--only', default='self,pts,img')
    ap.add_argument('--no-center', action='store_true')
    ap.add_argument('--f32-offlog', action='store_true')
    ap.add_argument('--img-hw', type=int, nargs=2, default=[256, 704], help='cat-128 config: 800 1440')
    ap.add_argument('--dh', type=int, default=32, help='channels per head (cat-128 config: 16)')
    a = ap.parse_args()
    dev = 'cuda'
    dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[a.dtype]
    for name in a.only.split(','):
        value, offlog, ref, vis0, count, gout, geom, is_grid, center = instance(
            name, a.bs, dtype, dev, not a.random_offsets, tuple(a.img_hw), a.dh)
        B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh = geom
        if dtype != torch.float32 and not a.f32_offlog:
            offlog = offlog.to(dtype)        # what the sampling_offsets Linear emits under autocast
        value.requires_grad_()
        offlog.requires_grad_()
        lists = UF.compact_visible(vis0, qw) if vis0 is not None else None
        for it in range(a.iters + 3):
            if it == 3:
                UF.kernel_profile(True)
            out = UF.bev_lift(value, offlog, ref, Nc, (fh, fw), H, P, vis0=vis0, count=count,
                              query_grid=(qh, qw), ref_is_grid=is_grid,
                              slot_center=None if a.no_center else center, visible_lists=lists)
            out.backward(gout)
            value.grad = offlog.grad = None
        res = UF.kernel_profile()
        UF.kernel_profile(False)
        tot = {'fwd': [0.0, 0.0], 'bwd': [0.0, 0.0]}
        for kname, r in sorted(res.items()):
            kind = 'fwd' if 'fwd' in kname else 'bwd'
            tot[kind][0] += r['avg_us']
            tot[kind][1] = max(tot[kind][1], r['bytes_per_launch'])
            gbs = r['bytes_per_launch'] / r['avg_us'] / 1e3
            print(f'  {kname:86s} {r["avg_us"]:9.1f} us  alg {r["bytes_per_launch"] / 1e6:7.1f} MB '
                  f'{gbs:8.1f} GB/s ({100 * gbs / 8000:5.2f}% of 8 TB/s)')
        print(f'{name:5s} B={B} {a.dtype}: fwd {tot["fwd"][0]:8.1f} us   bwd (all kernels) {tot["bwd"][0]:8.1f} us')


if __name__ == '__main__':
    main()
