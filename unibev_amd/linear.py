"""Linear layers of the BEV encoder with a split-K weight gradient.

Forward and the input gradient are plain library GEMMs (M = bs*40 000 rows, hipBLASLt through
torch).  The weight gradient dW = dY^T X reduces over those M rows into a 256x256 .. 512x256
output: as one GEMM it occupies (N/64)*(K/64) = 16..32 workgroups of the 256 CUs (measured 198 us
per call, 5.9 ms per training step, profiles/r01_v0_*).  Here the rows are split into S slices, the
slices run as one strided-batched GEMM that fills the chip, and the S partial products are summed.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function


def _splits(rows):
    for s in (32, 16, 8, 4, 2):
        if rows % s == 0 and rows // s >= 256:
            return s
    return 1


class _LinearSplitK(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        go2 = grad_out.reshape(-1, grad_out.shape[-1])
        if ctx.needs_input_grad[0]:
            gx = (go2 @ weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, x.shape[-1])
            rows = x2.shape[0]
            s = _splits(rows)
            if s > 1:
                part = torch.bmm(go2.view(s, rows // s, -1).transpose(1, 2), x2.view(s, rows // s, -1))
                gw = part.sum(0, dtype=torch.float32).to(weight.dtype)
            else:
                gw = go2.t() @ x2
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = go2.sum(0, dtype=torch.float32).to(grad_out.dtype)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """``F.linear`` with the split-K weight gradient; follows the ambient autocast dtype."""
    if x.is_cuda and torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        x, weight = x.to(dt), weight.to(dt)
        bias = None if bias is None else bias.to(dt)
        with torch.autocast('cuda', enabled=False):
            return _LinearSplitK.apply(x, weight, bias)
    return _LinearSplitK.apply(x, weight, bias)
