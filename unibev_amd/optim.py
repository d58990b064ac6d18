"""``FlatAdamW``: gradient clipping + AdamW over flat f32 buffers (``ubv_sumsq_f32`` + ``ubv_adamw_flat``).

Semantics of the reference's training step — mmcv ``OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))``
followed by ``torch.optim.AdamW`` (projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:
455-462) — as two streaming passes over ONE buffer each of parameters, gradients and moments instead of ~12
multi-tensor launches over ~200 tensors (0.4 ms per step at the encoder's 13.9 M parameters).

The parameters are re-pointed at views of the flat buffer (``p.data``); the gradients come from a
``dp.FlatGradients`` (the buffer the gradient all-reduce already uses).

Parameter groups: ``params`` is a list of tensors (one lr / weight decay) or torch-style groups
``[{'params': [...], 'lr': ..., 'weight_decay': ...}, ...]``; ``paramwise_groups`` makes them from the reference's
``paramwise_cfg = dict(custom_keys = {'img_backbone': dict(lr_mult = 0.1)})`` (config :455-462) the way mmcv's
``DefaultOptimizerConstructor`` does.  The groups stay ONE pass over the flat buffers: runs of consecutive parameters with
the same (lr, weight decay) become ranges of ``ubv_adamw_flat_groups`` (one step counter, one clip coefficient over all
groups, as in one torch optimizer).

Restriction, enforced loudly: EVERY parameter must receive a gradient in every step: ``torch.optim.AdamW`` skips
a parameter whose grad is None (no decay, no moment update), a flat pass cannot, so ``step`` raises when the gradient
collection reports parameters without a gradient (freeze them with ``requires_grad_(False)`` and leave them out, or
use the torch optimizer: ``bench.py --torch-optimizer``).
"""
import ctypes

import torch

from . import functional as UF
from . import linear as UL
from ._lib import lib, check


def paramwise_groups(named_params, lr, weight_decay, custom_keys=None, bias_lr_mult=1.0, bias_decay_mult=1.0,
                     norm_decay_mult=1.0, norm_names=(), **unsupported):
    """torch-style parameter groups, one per parameter, from mmcv's ``paramwise_cfg`` (mmcv
    ``DefaultOptimizerConstructor.add_params``; the reference's configs use ``custom_keys`` only, :455-462): a
    parameter whose name contains a custom key takes that key's ``lr_mult`` / ``decay_mult`` — the LONGEST matching
    key wins, ties in alphabetical order; otherwise ``bias_lr_mult`` / ``bias_decay_mult`` apply to parameters named
    ``bias`` and ``norm_decay_mult`` to parameters of the modules listed in ``norm_names`` (name prefixes of the
    normalisation layers: the caller knows its modules, this function sees names only).  As in the constructor,
    ``bias_lr_mult`` does NOT apply to a normalisation layer's bias, and a normalisation layer's parameters take
    ``norm_decay_mult`` only.  Frozen parameters are left out, like the constructor leaves them without a step.
    ``dwconv_decay_mult`` / ``dcn_offset_lr_mult`` (depth-wise convolutions, DCN offset convolutions: module types this
    function cannot see from names) are rejected instead of ignored; no shipped config sets them."""
    if unsupported:
        raise ValueError(f'paramwise_groups: unsupported paramwise_cfg keys {sorted(unsupported)} (supported: custom_keys, '
                         f'bias_lr_mult, bias_decay_mult, norm_decay_mult)')
    keys = sorted(sorted((custom_keys or {}).keys()), key=len, reverse=True)
    groups = []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        g = {'params': [p], 'lr': float(lr), 'weight_decay': float(weight_decay)}
        for k in keys:
            if k in name:
                g['lr'] = float(lr) * float(custom_keys[k].get('lr_mult', 1.0))
                g['weight_decay'] = float(weight_decay) * float(custom_keys[k].get('decay_mult', 1.0))
                break
        else:
            is_norm = any(name.startswith(n + '.') for n in norm_names)
            if (name.endswith('.bias') or name == 'bias') and not is_norm:
                g['lr'] = float(lr) * float(bias_lr_mult)
                g['weight_decay'] = float(weight_decay) * float(bias_decay_mult)
            if is_norm:
                g['weight_decay'] = float(weight_decay) * float(norm_decay_mult)
        groups.append(g)
    return groups


def group_ranges(groups, lr, weight_decay):
    """(parameters in order, [(end element, lr, weight decay), ...]): runs of consecutive parameters with one
    (lr, weight decay) merged into one range of the flat buffers."""
    params, ranges, o = [], [], 0
    for g in groups:
        glr, gwd = float(g.get('lr', lr)), float(g.get('weight_decay', weight_decay))
        for p in g['params']:
            params.append(p)
            o += p.numel()
            if ranges and ranges[-1][1:] == (glr, gwd):
                ranges[-1] = (o, glr, gwd)
            else:
                ranges.append((o, glr, gwd))
    return params, ranges


class FlatAdamW:
    def __init__(self, params, grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        params = list(params)
        if params and isinstance(params[0], dict):
            self.params, self.ranges = group_ranges(params, lr, weight_decay)
        else:
            self.params, self.ranges = params, [(sum(p.numel() for p in params), float(lr), float(weight_decay))]
        assert [id(p) for p in self.params] == [id(p) for p in grads.params], 'same parameters, same order'
        assert all(p.dtype == torch.float32 and p.is_cuda for p in self.params)
        if len(self.ranges) > int(lib().ubv_adamw_flat_max_groups()):
            raise ValueError(f'FlatAdamW: {len(self.ranges)} runs of parameters with their own lr / weight decay, at most '
                             f'{int(lib().ubv_adamw_flat_max_groups())}: order the parameters group by group')
        self.grads = grads
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self._base = self._initial = (float(lr), float(weight_decay))
        k = len(self.ranges)
        self._ends = (ctypes.c_int64 * k)(*[r[0] for r in self.ranges])
        self._lrs, self._wds = (ctypes.c_float * k)(), (ctypes.c_float * k)()
        # torch-style mutable settings, one dict per range of the flat buffers, READ AT EVERY STEP: a scheduler writes
        # ``group['lr']`` (mmcv's LrUpdaterHook computes it from ``group['initial_lr']``; the reference's schedule is
        # linear warm-up + CosineAnnealing, a new lr every iteration: config :463-469) or ``opt.lr = x`` (below)
        self.param_groups = [{'lr': r[1], 'initial_lr': r[1], 'weight_decay': r[2], 'initial_weight_decay': r[2],
                              'range': (0 if i == 0 else self.ranges[i - 1][0], r[0])} for i, r in enumerate(self.ranges)]
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

    @property
    def lr(self):
        """The base learning rate.  Assigning scales every group: ``group['lr'] = x * initial_lr / initial base lr`` (the
        groups keep their ``lr_mult``)."""
        return self._base[0]

    @lr.setter
    def lr(self, x):
        b0 = self.param_groups[0]['initial_lr'] if not self._initial[0] else self._initial[0]
        for g in self.param_groups:
            g['lr'] = float(x) * (g['initial_lr'] / b0 if b0 else 1.0)
        self._base = (float(x), self._base[1])

    @property
    def weight_decay(self):
        return self._base[1]

    @weight_decay.setter
    def weight_decay(self, x):
        b0 = self._initial[1]
        for g in self.param_groups:
            g['weight_decay'] = float(x) * (g['initial_weight_decay'] / b0 if b0 else 1.0)
        self._base = (self._base[0], float(x))

    @torch.no_grad()
    def step(self):
        """One update from ``grads.flat`` (the parameters' ``.grad`` views).  Nothing is read back."""
        g = self.grads.flat
        if self.grads.missing:
            names = ', '.join(str(tuple(self.params[i].shape)) for i in self.grads.missing[:4])
            raise RuntimeError(f'FlatAdamW: {len(self.grads.missing)} parameter(s) received no gradient in this step '
                               f'(shapes {names} ...): torch.optim.AdamW would skip them, the flat pass would decay '
                               f'them.  Exclude them (requires_grad_(False)) or use the torch optimizer.')
        for i, grp in enumerate(self.param_groups):     # (k <= 16 floats: the host arrays are by-value kernel arguments)
            self._lrs[i], self._wds[i] = float(grp['lr']), float(grp['weight_decay'])
        with UF._need_cuda(self.flat, g):
            st = UF._stream()
            sq = None
            if self.max_grad_norm is not None:
                check(lib().ubv_sumsq_f32(UF._p(g), g.numel(), UF._p(self.sumsq), UF._p(self._ws), st), 'sumsq_f32')
                sq = self.sumsq
            check(lib().ubv_adamw_flat_groups(UF._p(self.flat), UF._p(g), UF._p(self.exp_avg), UF._p(self.exp_avg_sq),
                                              g.numel(), len(self.ranges), self._ends, self._lrs, self._wds,
                                              self.betas[0], self.betas[1], self.eps, UF._p(self.step_count),
                                              UF._p(sq), self.max_grad_norm or 0.0, st), 'adamw_flat_groups')
        UL.mark_weights_changed()                       # the 16-bit / split weight copies are stale

    def grad_norm(self):
        """The gradient norm the last ``step`` clipped against (reads one float back)."""
        return float(self.sumsq.sqrt().item())

    def zero_grad(self):
        self.grads.flat.zero_()
