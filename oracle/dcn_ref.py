"""CPU oracle of the modulated deformable convolution (DCNv2) — TEST INFRASTRUCTURE ONLY.

Restates [ext] mmcv 1.4.0 ``modulated_deform_conv`` (ops/csrc/common/cuda/modulated_deform_conv_cuda_kernel.cuh:
``dmcn_im2col_bilinear`` + ``modulated_deformable_im2col_gpu_kernel``, and ``ModulatedDeformConv2dPack.forward`` in
ops/modulated_deform_conv.py) with differentiable torch ops, so autograd supplies the reference gradients.
mmcv is a dependency that is NOT in /root/reference (requirements: mmcv-full 1.4.0): PARITY UNPINNED — the
restatement follows the published kernel; known-answer checks (zero offsets + unit mask = ``F.conv2d``) anchor it
in tests/test_oracle_dcn.py.

    position (tap i, j of output pixel ho, wo) = (ho*s - pad + i*dil + dy, wo*s - pad + j*dil + dx)
    value = 0 when the position is <= -1 or >= size; otherwise bilinear over the 4 corners, corners outside the
    map reading 0.   offset[:, 2*(g*K + k)] = dy, [:, 2*(g*K + k) + 1] = dx, mask[:, g*K + k] = modulation.
"""
import torch
import torch.nn.functional as F


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                            deform_groups=1):
    assert groups == 1
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    N, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    K, dg, Cg = kh * kw, deform_groups, C // deform_groups
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    ho = torch.arange(Ho, dtype=x.dtype).view(1, 1, Ho, 1)
    wo = torch.arange(Wo, dtype=x.dtype).view(1, 1, 1, Wo)
    cols = []
    for k in range(K):
        i, j = divmod(k, kw)
        per_group = []
        for g in range(dg):
            dy = offset[:, 2 * (g * K + k)].unsqueeze(1)                 # [N, 1, Ho, Wo]
            dx = offset[:, 2 * (g * K + k) + 1].unsqueeze(1)
            hp = ho * sh - ph + i * dh + dy
            wp = wo * sw - pw + j * dw + dx
            live = ((hp > -1) & (wp > -1) & (hp < H) & (wp < W)).to(x.dtype)
            h0, w0 = torch.floor(hp), torch.floor(wp)
            lh, lw = hp - h0, wp - w0
            xg = x[:, g * Cg:(g + 1) * Cg].reshape(N, Cg, H * W)
            val = 0
            for (hc, wc, wt) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                 (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                ok = ((hc >= 0) & (wc >= 0) & (hc <= H - 1) & (wc <= W - 1)).to(x.dtype) * live
                idx = (hc.clamp(0, H - 1) * W + wc.clamp(0, W - 1)).long().view(N, 1, Ho * Wo).expand(N, Cg, Ho * Wo)
                v = torch.gather(xg, 2, idx).view(N, Cg, Ho, Wo)
                val = val + v * (wt * ok)
            per_group.append(val * mask[:, g * K + k].unsqueeze(1))
        cols.append(torch.cat(per_group, 1))                              # [N, C, Ho, Wo]
    col = torch.stack(cols, 2)                                            # [N, C, K, Ho, Wo]
    out = torch.einsum('nckhw,ock->nohw', col, weight.reshape(Cout, C, K))
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def dcn_pack(x, conv_offset_weight, conv_offset_bias, weight, bias, stride, padding, dilation, deform_groups):
    """``ModulatedDeformConv2dPack.forward``: offsets and mask from a plain convolution of the same geometry."""
    out = F.conv2d(x, conv_offset_weight, conv_offset_bias, stride, padding, dilation)
    o1, o2, m = torch.chunk(out, 3, dim=1)
    return modulated_deform_conv2d(x, torch.cat((o1, o2), 1), torch.sigmoid(m), weight, bias, stride, padding,
                                   dilation, 1, deform_groups)
