#!/usr/bin/env python3
"""Row f4 timings: the DCNv2 operator at the ResNet-101 stage-3 / stage-4 shapes of the 256x704 config (12 images)
next to a plain MIOpen convolution of the same geometry, and forward + backward of the two backbone / neck pairs."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF      # noqa: E402
from unibev_amd.modules import FPN, SECOND, SECONDFPN, ResNet, extract_img_feat      # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def dcn_op(dtype):
    for (N, C, H, W, Cout) in ((12, 256, 16, 44, 256), (12, 512, 8, 22, 512)):
        x = torch.randn(N, C, H, W, device='cuda').to(dtype).to(memory_format=torch.channels_last).requires_grad_()
        off = (1.5 * torch.randn(N, 18, H, W, device='cuda')).to(dtype).requires_grad_()
        m = torch.rand(N, 9, H, W, device='cuda').to(dtype).requires_grad_()
        w = (torch.randn(Cout, C, 3, 3, device='cuda') / (3 * C ** 0.5)).requires_grad_()
        gy = torch.randn(N, Cout, H, W, device='cuda').to(dtype).to(memory_format=torch.channels_last)

        def fwd():
            return UF.modulated_deform_conv2d(x, off, m, w, None, 1, 1, 1, 1, 1)

        def both():
            y = fwd()
            y.backward(gy)
            x.grad = off.grad = m.grad = w.grad = None

        wl = w.detach().to(dtype).to(memory_format=torch.channels_last).requires_grad_()

        def lib_both():
            y = torch.nn.functional.conv2d(x, wl, None, 1, 1)
            y.backward(gy)
            x.grad = wl.grad = None

        e = x.element_size()
        comp = (N * H * W * (C + Cout) + N * 27 * H * W) * e          # feature map in, result out, offsets + mask
        tf, tb, tl = timeit(fwd), timeit(both), timeit(lib_both)
        flops = 2.0 * N * H * W * Cout * C * 9
        print(f'dcn {str(dtype):15s} N={N} C={C} {H}x{W} -> {Cout}: fwd {tf:7.1f} us ({comp / tf / 1e3:6.0f} GB/s of '
              f'compulsory bytes, {flops / tf / 1e6:6.1f} TFLOP/s)  fwd+bwd {tb:7.1f} us   '
              f'[plain conv2d fwd+bwd, MIOpen: {tl:7.1f} us]')


def nets(dtype, cams, bs):
    ac = torch.autocast('cuda', dtype=dtype) if dtype != torch.float32 else torch.autocast('cuda', enabled=False)
    img_b = ResNet(depth=101, num_stages=4, out_indices=(3,), frozen_stages=1,
                   norm_cfg=dict(type='BN2d', requires_grad=False), norm_eval=True, style='caffe', with_cp=True,
                   dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False),
                   stage_with_dcn=(False, False, True, True)).cuda().train().to(memory_format=torch.channels_last)
    img_n = FPN(in_channels=[2048], out_channels=256, start_level=0, add_extra_convs='on_output', num_outs=1,
                relu_before_extra_convs=True).cuda()
    img = torch.randn(bs, cams, 3, 256, 704, device='cuda')

    def img_step():
        with ac:
            f = extract_img_feat(img, img_b, img_n)[0]
        f.float().square().mean().backward()
        for p in list(img_b.parameters()) + list(img_n.parameters()):
            p.grad = None

    t0 = time.time()
    t = timeit(img_step, n=3, warm=2)
    print(f'img backbone + neck ({str(dtype)}, {bs * cams} x 3x256x704, ResNet-101 DCNv2 + FPN, checkpointed) fwd+bwd '
          f'{t / 1e3:8.2f} ms  ({time.time() - t0:.0f} s incl. MIOpen search)')
    pts_b = SECOND(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2]).cuda().train()
    pts_n = SECONDFPN(in_channels=[128, 256], upsample_strides=[1, 2], out_channels=[128, 128],
                      use_conv_for_no_stride=True).cuda().train()
    x = torch.randn(bs, 256, 180, 180, device='cuda', requires_grad=True)

    def pts_step():
        with ac:
            f = pts_n(pts_b(x))[0]
        f.float().square().mean().backward()
        for p in list(pts_b.parameters()) + list(pts_n.parameters()):
            p.grad = None

    t = timeit(pts_step, n=5, warm=2)
    print(f'pts backbone + neck ({str(dtype)}, {bs} x 256x180x180, SECOND + SECONDFPN) fwd+bwd {t / 1e3:8.2f} ms')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-nets', action='store_true')
    ap.add_argument('--bs', type=int, default=2)
    a = ap.parse_args()
    for dt in (torch.float32, torch.bfloat16):
        dcn_op(dt)
    if not a.no_nets:
        for dt in (torch.bfloat16, torch.float32):
            nets(dt, 6, a.bs)
