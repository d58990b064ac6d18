"""``FlatAdamW``: gradient clipping + AdamW over flat f32 buffers (``ubv_sumsq_f32`` + ``ubv_adamw_flat``).

Semantics of the reference's training step — mmcv ``OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))``
followed by ``torch.optim.AdamW`` (projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:
455-462) — as two streaming passes over ONE buffer each of parameters, gradients and moments instead of ~12
multi-tensor launches over ~200 tensors (0.4 ms per step at the encoder's 13.9 M parameters).

The parameters are re-pointed at views of the flat buffer (``p.data``); the gradients come from a
``dp.FlatGradients`` (the buffer the gradient all-reduce already uses).

Restrictions, enforced loudly: ONE parameter group (one lr / weight decay — what the encoder-side parameters of the
shipped configs share; the configs' ``paramwise_cfg`` lr_mult = 0.1 applies to ``img_backbone`` only, which is outside
this optimizer's parameter list), and EVERY parameter must receive a gradient in every step: ``torch.optim.AdamW`` skips
a parameter whose grad is None (no decay, no moment update), a flat pass cannot, so ``step`` raises when the gradient
collection reports parameters without a gradient (freeze them with ``requires_grad_(False)`` and leave them out, or
use the torch optimizer: ``bench.py --torch-optimizer``).
"""
import torch

from . import functional as UF
from . import linear as UL
from ._lib import lib, check


class FlatAdamW:
    def __init__(self, params, grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        self.params = list(params)
        assert [id(p) for p in self.params] == [id(p) for p in grads.params], 'same parameters, same order'
        assert all(p.dtype == torch.float32 and p.is_cuda for p in self.params)
        self.grads = grads
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), \
            float(weight_decay)
        self.max_grad_norm = None if max_grad_norm is None else float(max_grad_norm)
        dev = self.params[0].device
        n = grads.flat.numel()
        with torch.no_grad():
            self.flat = torch.cat([p.detach().reshape(-1) for p in self.params])
            o = 0
            for p in self.params:                       # the parameters now live in the flat buffer
                p.data = self.flat[o:o + p.numel()].view_as(p)
                o += p.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = torch.empty(int(lib().ubv_sumsq_workspace()), dtype=torch.uint8, device=dev)
        UL.mark_weights_changed()                       # cached low-precision copies point at the old storage

    @torch.no_grad()
    def step(self):
        """One update from ``grads.flat`` (the parameters' ``.grad`` views).  Nothing is read back."""
        g = self.grads.flat
        if self.grads.missing:
            names = ', '.join(str(tuple(self.params[i].shape)) for i in self.grads.missing[:4])
            raise RuntimeError(f'FlatAdamW: {len(self.grads.missing)} parameter(s) received no gradient in this step '
                               f'(shapes {names} ...): torch.optim.AdamW would skip them, the flat pass would decay '
                               f'them.  Exclude them (requires_grad_(False)) or use the torch optimizer.')
        with UF._need_cuda(self.flat, g):
            st = UF._stream()
            sq = None
            if self.max_grad_norm is not None:
                check(lib().ubv_sumsq_f32(UF._p(g), g.numel(), UF._p(self.sumsq), UF._p(self._ws), st), 'sumsq_f32')
                sq = self.sumsq
            check(lib().ubv_adamw_flat(UF._p(self.flat), UF._p(g), UF._p(self.exp_avg), UF._p(self.exp_avg_sq),
                                       g.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                       UF._p(self.step_count), UF._p(sq), self.max_grad_norm or 0.0, st), 'adamw_flat')
        UL.mark_weights_changed()                       # the 16-bit / split weight copies are stale

    def grad_norm(self):
        """The gradient norm the last ``step`` clipped against (reads one float back)."""
        return float(self.sumsq.sqrt().item())

    def zero_grad(self):
        self.grads.flat.zero_()
