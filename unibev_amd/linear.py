"""Linear layers of the BEV encoder: library GEMMs with a split-K weight gradient and per-step cached
low-precision weights.

* Forward and the input gradient are plain library GEMMs (M = bs*40 000 rows, hipBLASLt via torch).
* The weight gradient dW = dY^T X reduces over those M rows into a 256x256 .. 512x256 output: as one
  GEMM it occupies (N/64)*(K/64) = 16..32 workgroups of the 256 CUs (measured 198 us per call,
  5.9 ms per training step, profiles/r01_v0_*).  Here the rows are split into S slices that run as
  one strided-batched GEMM filling the chip, and the S partial products are summed in f32.
* The partial products and the bias column sums are reduced by one library launch per Linear
  (``ubv_linear_grad_reduce``); when the Linear's output went straight into the fused
  add+dropout+LayerNorm, that kernel's backward supplies the bias gradient and grad_out is not read
  again.
* Under autocast the f32 master weights are cast to the autocast dtype by ONE multi-tensor copy
  (``lowp_step_cache``) after each optimizer step, not by one cast kernel per use, and the
  gradients come back in f32 directly: per step this removes ~300 tiny cast kernels and their
  autograd nodes (the eager step is host-bound below ~3.5 us per kernel, MI355X_MICROARCH.md).
* ``linear_cat`` runs several Linear layers that share their input as ONE GEMM (the
  ``sampling_offsets`` and ``attention_weights`` layers of every deformable attention).
* ``linear_pass`` / ``linear_cat_pass`` also return an alias of the input for the caller's residual
  branch, so that the residual's gradient is accumulated by the input-gradient GEMM itself.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.optim.optimizer import register_optimizer_step_post_hook

import contextlib
import os

# (param ids, dtype) -> (concatenated low-precision buffer, per-parameter views, parameters)
_SHADOWS = {}
_ACTIVE = False


def _splits(rows):
    for s in (32, 16, 8, 4, 2):
        if rows % s == 0 and rows // s >= 256:
            return s
    return 1


# Shadows are refreshed when the masters may have changed: after any optimizer step (a global
# post-step hook; the fused AdamW kernel does not bump parameter versions) or when a parameter's
# version counter moved (load_state_dict, in-place edits).  Forward passes in between — gradient
# accumulation, several forwards before one backward — reuse the shadows untouched, so the copies
# saved for backward stay valid.
_DIRTY = [True]
_PLAN = {'n': -1, 'groups': [], 'params': [], 'versions': None}   # cached refresh plan


def _mark_dirty(*_args, **_kwargs):
    _DIRTY[0] = True


register_optimizer_step_post_hook(_mark_dirty)


def mark_weights_changed():
    """Call after changing parameters in a way neither an optimizer step nor the version counters
    reveal (e.g. through ``param.data``)."""
    _DIRTY[0] = True


def _refresh_plan():
    """(dst views, src parameters) per (dtypes, device) group; rebuilt only when shadows were added."""
    if _PLAN['n'] != len(_SHADOWS):
        groups, flat = {}, []
        for buf, views, params in _SHADOWS.values():
            g = groups.setdefault((buf.dtype, params[0].dtype, buf.device), ([], []))
            g[0].extend(views)
            g[1].extend(params)
            flat.extend(params)
        _PLAN.update(n=len(_SHADOWS), groups=list(groups.values()), params=flat, versions=None)
    return _PLAN


def _masters_changed():
    if _DIRTY[0]:
        return True
    plan = _refresh_plan()
    return plan['versions'] != [p._version for p in plan['params']]


@contextlib.contextmanager
def lowp_step_cache():
    """Within this context (one forward pass of the encoder) the low-precision copies of the
    weights are taken from persistent shadow buffers, refreshed ON ENTRY with a single multi-tensor
    copy (``torch._foreach_copy_``) whenever the f32 masters may have changed since the last
    refresh, instead of one cast kernel per use."""
    global _ACTIVE
    if _ACTIVE:                       # nested: the outer context already refreshed
        yield
        return
    from . import functional as UF
    UF.new_step()                     # backward accumulators of this pass come from a fresh arena
    _SPLIT_PASS.clear()
    _presplit()
    if _SHADOWS and _masters_changed():
        plan = _refresh_plan()
        with torch.no_grad():
            for dst, src in plan['groups']:
                torch._foreach_copy_(dst, src)
        plan['versions'] = [p._version for p in plan['params']]
        _DIRTY[0] = False
    _ACTIVE = True
    try:
        yield
    finally:
        _ACTIVE = False


def clear_lowp_cache():
    _SPLIT_USED.clear()
    _SPLIT_PREV.clear()
    _SHADOWS.clear()
    _PLAN.update(n=-1, groups=[], params=[], versions=None)
    _DIRTY[0] = True


# f32 Linear layers on the matrix cores: per forward pass, the weight of a Linear is split once into
# bf16 halves (both orientations) for functional.gemm_nt; fresh buffers per pass, so the copies a
# pass saved for its backward stay valid when a later pass starts (gradient accumulation).
_SPLIT_PASS = {}
_MFMA_F32 = os.environ.get('UBV_F32_GEMM', 'mfma') != 'library'


def set_f32_gemm(mode):
    """'mfma': f32 Linear layers as split-bf16 products on the matrix cores (default; ~2^-17 per
    product, BEV features 3e-4 from the reference's at full size).  'library': IEEE f32 GEMMs
    (hipBLASLt; 6e-5), 1.7x slower.  Returns the previous mode."""
    global _MFMA_F32
    if mode not in ('mfma', 'library'):
        raise ValueError("mode must be 'mfma' or 'library'")
    prev = 'mfma' if _MFMA_F32 else 'library'
    _MFMA_F32 = mode == 'mfma'
    return prev
_MFMA_WGRAD = os.environ.get('UBV_WGRAD', 'mfma') != 'library'
# 16-bit data: every Linear (forward and input gradient) on this library's MFMA kernel (round 6: the 16-bit step held 60
# hipBLASLt `Cijk_*` launches, a sixth of its kernel time).  UBV_MFMA16_MAXN=192 restores round 5's split —
# the MFMA kernel for narrow outputs, the library for wide ones (A/B: profiles/r06_gemm16.txt)
_MFMA_16_MAXN = int(os.environ.get('UBV_MFMA16_MAXN', '4096'))
_WT16 = {}                   # id(16-bit weight buffer) -> (version, transposed copy): the input gradient's operand


def _transposed16(w):
    """w^T [K, N] contiguous for dX = dY . W as ``gemm_nt(dY, w^T)``; one copy per refresh of the shadow weights (their
    buffers are updated in place: the version counter tells)."""
    hit = _WT16.get(id(w))
    if hit is not None and hit[0] == w._version and hit[1].device == w.device and hit[2] is w:
        return hit[1]
    if len(_WT16) > 1024:
        _WT16.clear()
    with torch.no_grad():
        wt = w.t().contiguous()
    _WT16[id(w)] = (w._version, wt, w)
    return wt


_SPLIT_USED = {}            # weight groups the CURRENT pass asked for: key -> parameters
_SPLIT_PREV = {}            # ... the previous pass asked for: split together, in one launch, when the next pass starts


def _presplit():
    """On entry to a pass: the weight groups the previous pass used are split in ONE launch
    (``split_weights_batched``: 46 launches of ~5 us each per f32 step otherwise).  A pass that asks for none (16-bit
    autocast) makes the next one start empty again."""
    global _SPLIT_USED, _SPLIT_PREV
    _SPLIT_PREV, _SPLIT_USED = _SPLIT_USED, {}
    if not _SPLIT_PREV or not _MFMA_F32:
        return
    groups = [g for g in _SPLIT_PREV.values()
              if all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 2 for p in g)]
    if not groups:
        return
    from . import functional as UF
    by_dev = {}
    for g in groups:
        by_dev.setdefault(g[0].device, []).append(g)
    for gs in by_dev.values():
        for g, out in zip(gs, UF.split_weights_batched(gs)):
            _SPLIT_PASS[tuple(id(p) for p in g)] = out


def _masters(params):
    """Aliases made by ``functional.fan_out`` stand for their parameter in the per-step caches."""
    return tuple(getattr(p, '_ubv_master', p) for p in params)


def _split_weights(weights):
    weights = _masters(weights)
    key = tuple(id(p) for p in weights)
    hit = _SPLIT_PASS.get(key) if _ACTIVE else None
    if hit is None:
        from . import functional as UF
        w = weights[0] if len(weights) == 1 else torch.cat([p.detach() for p in weights], 0)
        hit = UF.split_weight(w)
        if _ACTIVE:
            _SPLIT_PASS[key] = hit
    if _ACTIVE:
        _SPLIT_USED[key] = list(weights)
    return hit


def _cached_lowp(params, dtype):
    """Concatenation (dim 0) of ``params`` in ``dtype``."""
    params = _masters(params)
    if all(p.dtype == dtype for p in params) and len(params) == 1:
        return params[0]
    key = tuple(id(p) for p in params) + (dtype,)
    if _ACTIVE:
        hit = _SHADOWS.get(key)
        if hit is not None and hit[0].device == params[0].device:
            return hit[0]
    with torch.no_grad():
        buf = torch.cat([p.detach() for p in params], 0).to(dtype) if len(params) > 1 \
            else params[0].detach().to(dtype)
    if _ACTIVE and all(isinstance(p, torch.nn.Parameter) for p in params):
        if len(_SHADOWS) > 4096:
            _SHADOWS.clear()
        views = list(torch.split(buf, [p.shape[0] for p in params], 0))
        _SHADOWS[key] = (buf, views, list(params))
        # the refresh plan holds the old buffers: rebuild it (a shadow REPLACED under its key, or the table
        # cleared and refilled to the same size, leaves len(_SHADOWS) unchanged)
        _PLAN['n'] = -1
        _PLAN['versions'] = None      # a shadow created mid-pass: its versions are taken at the next refresh
    return buf


def _row_bias_grad(g3, slot):
    """d(row_bias) = the sum of grad_out's row blocks g3 [S, R, N] over the samples.  ``slot`` = (G, col0) when the
    row bias is a column slice of a positional fold's output (``_PosFoldAll``): the sum is written straight into the same
    columns of the fold's gradient matrix G and that slice is returned — the fold's backward then finds all its gradients
    in ONE matrix.  This library's kernels only (plain f32 adds: the two-stream window, unibev_amd/debug.py)."""
    from . import functional as UF
    if slot is not None and g3.is_cuda and g3.dtype == torch.float32 and g3.is_contiguous() and g3.shape[2] % 4 == 0:
        G, c0 = slot
        if G.shape[0] == g3.shape[1] and c0 + g3.shape[2] <= G.shape[1]:
            return UF.slice_sum(g3, G[:, c0:c0 + g3.shape[2]])
    if g3.shape[0] == 1:
        return g3[0]
    if g3.dtype == torch.float32 and g3.is_cuda and g3.is_contiguous() and g3[0].numel() % 4 == 0:
        return UF.linear_grad_reduce(None, g3)[1]
    grb = g3[0] + g3[1]
    for i in range(2, g3.shape[0]):
        grb = grb + g3[i]
    return grb


class _PosFoldAll(Function):
    """The positional terms of ALL layers of an encoder — ``bev_pos . [W_so; W_aw]_l^T`` for every layer l — as ONE GEMM
    over the (Nq, C) table, returned as per-layer column views of its output; backward: the layers wrote their gradients
    into the matching columns of ONE matrix (``_row_bias_grad``), so the table's gradient is one input-gradient GEMM and
    the weights' one weight-gradient pass.  Rounds 3 - 5 had the same GEMMs around ``torch.split``, whose backward is a
    framework cat (+ a zero fill per unused piece) inside the encoders' two-stream window."""

    @staticmethod
    def forward(ctx, base, *ws):
        from . import functional as UF
        split = _split_weights(ws)
        terms = UF.gemm_nt(base, split[0], split[1])
        if terms is None:
            raise RuntimeError('pos_fold_all: shape outside ubv_gemm_nt')
        ctx.save_for_backward(base, split[2], split[3])
        ctx.sizes = [w.shape[0] for w in ws]
        ctx.G = torch.empty_like(terms)                     # the layers' gradients, column block by column block
        ctx.set_materialize_grads(False)
        outs, c = [], 0
        for i in range(0, len(ws), 2):
            n = ws[i].shape[0] + ws[i + 1].shape[0]
            outs.append(terms[:, c:c + n])
            c += n
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        from . import functional as UF
        base, wth, wtl = ctx.saved_tensors
        G, c = ctx.G, 0
        for g in gs:                                        # (normally every g IS its column block of G already)
            n = G.shape[1] // len(gs) if g is None else g.shape[1]
            blk = G[:, c:c + n]
            if g is None:
                blk.zero_()
            elif g.data_ptr() != blk.data_ptr() or g.stride() != blk.stride():
                blk.copy_(g)
            c += n
        gx = UF.gemm_nt(G, wth, wtl) if ctx.needs_input_grad[0] else None
        res = UF.gemm_wgrad(G, base)
        if res is None or (ctx.needs_input_grad[0] and gx is None):
            raise RuntimeError('pos_fold_all: gradient shape outside the MFMA kernels')
        gw, grads, off = res[0], [], 0
        for n in ctx.sizes:
            grads.append(gw[off:off + n])
            off += n
        return (gx, *grads)


def pos_fold_all(base, ws):
    """Per-layer positional terms [(Nq, n_l)] for ws = [W_so_0, W_aw_0, W_so_1, ...] (f32 CUDA); each term carries its
    gradient slot (``_ubv_grad_slot``) for the consumer's backward."""
    outs = _PosFoldAll.apply(base, *ws)
    G = outs[0].grad_fn.G if outs[0].grad_fn is not None and hasattr(outs[0].grad_fn, 'G') else None
    c = 0
    for t in outs:
        if G is not None:
            t._ubv_grad_slot = (G, c)
        c += t.shape[1]
    return list(outs)


class _Linear(Function):
    """y = x @ cat(weights)^T + cat(biases).  ``n`` weights followed by ``n`` biases (or none).

    ``passthru``: also return an alias of ``x`` for the caller's residual branch.  The two
    gradients that autograd would otherwise add with a separate elementwise kernel then arrive
    together, and the input-gradient GEMM takes the alias' gradient as its accumulator input
    (``addmm``: D = grad_alias + grad_out @ W in the GEMM epilogue)."""

    @staticmethod
    def forward(ctx, x, dtype, n, has_bias, passthru, act, row_bias, *params):
        # row_bias [R, N] (or None): + row_bias[m % R] on row m of the (M, N) output — a term that is the same for
        # every sample of the batch (the Linear of ``query + query_pos`` split into query . W^T + (pos . W^T)[q])
        weights, biases = params[:n], params[n:]
        ctx.act = act
        # an unused output (the pass-through alias of the last Linear of a chain) must not come back as a zero-filled
        # gradient: that is a framework fill of the whole activation (66 MB for the LiDAR map) inside the backward
        ctx.set_materialize_grads(False)
        ctx.rb_rows = None if row_bias is None else row_bias.shape[0]
        ctx.rb_slot = None if row_bias is None else getattr(row_bias, '_ubv_grad_slot', None)
        w = _cached_lowp(weights, dtype)
        b = _cached_lowp(biases, dtype) if has_bias else None
        xc = x.to(dtype)
        if xc.is_cuda and not xc.is_contiguous():
            # e.g. the first layer's queries: the BEV embedding expanded over the batch (stride 0).  Materialised once here,
            # the MFMA kernels below take it; left as it was, every GEMM of that layer fell to the library path
            # (clone + sgemm + bias add: 116 us against 48 us at 80 000 x 256 x 256)
            xc = xc.contiguous()
        ctx.meta = (x.dtype, n, has_bias, [p.shape[0] for p in weights], [p.dtype for p in params])
        y = None
        split = None
        ok = xc.is_cuda and xc.is_contiguous() and xc.numel() > 0 and xc.shape[-1] % 32 == 0 and \
            w.shape[0] % 32 == 0
        from . import functional as UF
        # act = ('relu_drop', p): this Linear's output goes through ReLU + Dropout(p) — applied in the
        # GEMM epilogue (ubv_gemm_nt_act), the pre-activation is never written; act = ('masked_in', p):
        # this Linear's INPUT is such an activation's output (see backward).
        fuse = act is not None and act[0] == 'relu_drop'
        seed = UF.next_dropout_seed() if (fuse and act[1] > 0) else 0
        fused = False
        if ok and dtype == torch.float32 and _MFMA_F32 and w.dtype == torch.float32:
            # split-bf16 product on the matrix cores (ubv_gemm_nt): HBM-bound where the f32 library
            # GEMM is MFMA-bound (61 vs 103 us at 80 000 x 256 x 256)
            split = _split_weights(weights)
            if fuse:
                y = UF.gemm_nt_act(xc, split[0], split[1], bias=b, act=1, p=act[1], seed=seed)
                fused = y is not None
            if y is None and row_bias is not None and row_bias.dtype == torch.float32:
                y = UF.gemm_nt(xc, split[0], split[1], bias=b, row_bias=row_bias)
                if y is not None:
                    row_bias = None                        # added in the epilogue
            if y is None:
                y = UF.gemm_nt(xc, split[0], split[1], bias=b)
        elif ok and dtype != torch.float32 and w.dtype == dtype and w.is_contiguous() and \
                (fuse or w.shape[0] <= _MFMA_16_MAXN):
            b32 = None
            if has_bias:                  # the kernel adds the bias in f32: hand it the master values
                b32 = biases[0].detach() if n == 1 else torch.cat([p.detach() for p in biases])
            if fuse:
                y = UF.gemm_nt_act(xc, w, bias=b32, act=1, p=act[1], seed=seed)
                fused = y is not None
            elif w.shape[0] <= _MFMA_16_MAXN:
                y = UF.gemm_nt(xc, w, bias=b32)
        if y is None and xc.is_cuda and xc.dtype == w.dtype and (b is None or b.dtype == w.dtype) and \
                xc.is_contiguous() and w.is_contiguous() and xc.numel() > 0:
            # hipBLASLt with a cached plan (ubv_linear_forward): same GEMM, a third of the host time
            y = UF.linear_forward(xc, w, b)
        if y is None:
            y = F.linear(xc, w, b)
        emu = os.environ.get('UBV_GEMM_EMU', '')
        if emu and xc.dtype == torch.float32 and not (emu.endswith('-keep-offlog') and n == 2):
            # precision study only: what a split-bf16 MFMA GEMM (3 products, f32 accumulation) would return
            def split_(t):
                hi = t.bfloat16().float()
                return hi, (t - hi).bfloat16().float()
            xh, xl = split_(xc)
            wh, wl = split_(w)
            y = F.linear(xh, wh, b) + F.linear(xh, wl) + F.linear(xl, wh)
        if row_bias is not None:                           # paths without the epilogue term
            R = row_bias.shape[0]
            y = (y.reshape(-1, R, y.shape[-1]) + row_bias.to(y.dtype)).reshape(y.shape)
        if fuse and not fused:
            y = UF.relu_dropout_raw(y, act[1], seed) if y.is_cuda else \
                F.dropout(torch.relu(y), act[1], training=act[1] > 0)
        ctx.n_wt = 2 if split is not None else 0
        keep = (xc, w) + ((split[2], split[3]) if split is not None else ()) + ((y,) if fuse else ())
        ctx.save_for_backward(*keep)
        return (y, x.view_as(x)) if passthru else y

    @staticmethod
    def backward(ctx, grad_out, grad_alias=None):
        if grad_out is None:
            # (set_materialize_grads(False)) only the pass-through alias was used: x's gradient is the alias' own, nothing
            # reaches the weights
            return (grad_alias, None, None, None, None, None, None, *([None] * len(ctx.meta[4])))
        xc, w = ctx.saved_tensors[:2]
        wt = ctx.saved_tensors[2:2 + ctx.n_wt]        # transposed split halves (f32 MFMA path) or ()
        x_dtype, n, has_bias, outs, pdt = ctx.meta
        act = ctx.act
        from . import functional as UF
        if act is not None and act[0] == 'relu_drop' and UF.grad_tag_stale(grad_out, '_ubv_masked'):
            raise RuntimeError('linear_relu_dropout: the activation fed something besides '
                               'linear_after_relu_dropout (its gradient arrived partly pre-multiplied)')
        if act is not None and act[0] == 'relu_drop' and getattr(ctx, 'ubv_folded', False) and \
                not UF.grad_tag(grad_out, '_ubv_masked'):
            # a linear_after_relu_dropout consumed this activation (it folds the derivative into its input gradient and
            # tags the result) yet the gradient arriving here is untagged: autograd summed it with another consumer's
            # gradient into a fresh tensor.  Applying the derivative to the sum would scale the folded part twice.
            raise RuntimeError('linear_relu_dropout: the activation was also used by something besides '
                               'linear_after_relu_dropout; its gradient arrived as a sum of pre-multiplied and raw parts')
        if act is not None and act[0] == 'relu_drop' and not UF.grad_tag(grad_out, '_ubv_masked'):
            # the consumer did not fold the activation's derivative into its input gradient
            a_out = ctx.saved_tensors[2 + ctx.n_wt]
            if a_out.is_cuda:
                grad_out = UF.relu_dropout_grad_raw(grad_out, a_out, act[1])
            else:
                grad_out = grad_out * (a_out != 0).to(grad_out.dtype) / (1.0 - act[1])
        go2 = grad_out.reshape(-1, grad_out.shape[-1])
        grb = None
        if ctx.rb_rows is not None and ctx.needs_input_grad[6]:
            # d(row_bias) = sum over the batch of grad_out's row blocks (plain adds: the framework's outer-dimension
            # reduction is slow on this shape, see bricks._ExpandBatch)
            grb = _row_bias_grad(go2.reshape(-1, ctx.rb_rows, go2.shape[-1]), ctx.rb_slot)
        gx = None
        if act is not None and act[0] == 'masked_in' and ctx.needs_input_grad[0]:
            assert grad_alias is None, 'linear_after_relu_dropout has no pass-through output'
            # dX = (dY . W) * relu'/keep mask of the activation that produced this Linear's input, in the
            # GEMM epilogue (ubv_gemm_nt_act, act 2); the producing Linear then takes the gradient as is
            a2 = xc.reshape(-1, xc.shape[-1])
            if go2.is_cuda and go2.is_contiguous() and go2.dtype == xc.dtype == x_dtype and grad_alias is None:
                if wt and go2.dtype == torch.float32:
                    gx = UF.gemm_nt_act(go2, wt[0], wt[1], act=2, mask=a2, p=act[1])
                elif go2.dtype != torch.float32 and w.dtype == go2.dtype:
                    gx = UF.gemm_nt_act(go2, _transposed16(w), act=2, mask=a2, p=act[1])
            if gx is None:
                g = (go2 @ w).to(x_dtype)
                gx = UF.relu_dropout_grad_raw(g, a2, act[1]) if g.is_cuda else \
                    g * (a2 != 0).to(g.dtype) / (1.0 - act[1])
            gx = gx.view(xc.shape)
            UF.tag_grad(gx, '_ubv_masked', True)
        if gx is None and ctx.needs_input_grad[0] and wt and go2.dtype == torch.float32 and x_dtype == torch.float32 and \
                go2.is_contiguous() and go2.shape[1] % 32 == 0 and xc.shape[-1] % 32 == 0:
            # dX = dY . W on the matrix cores, the residual branch's gradient added in the epilogue
            # (in place when that tensor was produced for this edge alone)
            ga = None
            if grad_alias is not None and grad_alias.dtype == torch.float32:
                ga = grad_alias.reshape(-1, xc.shape[-1])
                ga = ga if ga.is_contiguous() else ga.contiguous()
            own = ga is not None and UF.grad_tag(grad_alias, '_ubv_owned') and \
                ga.data_ptr() == grad_alias.data_ptr()
            gx = UF.gemm_nt(go2, wt[0], wt[1], residual=ga, out=ga if own else None)
            if gx is not None:
                gx = gx.view(xc.shape)
                if grad_alias is not None and ga is None:
                    gx = gx + grad_alias.to(x_dtype)
        if gx is None and ctx.needs_input_grad[0] and go2.is_cuda and go2.dtype != torch.float32 and \
                w.dtype == go2.dtype == x_dtype and go2.is_contiguous() and w.is_contiguous() and \
                go2.shape[1] % 32 == 0 and xc.shape[-1] % 32 == 0 and xc.shape[-1] <= _MFMA_16_MAXN:
            # 16-bit dX = dY . W on this library's MFMA kernel (w^T cached per weight refresh), the residual branch's
            # gradient added in the epilogue — in place when that tensor was produced for this edge alone
            ga = None
            if grad_alias is not None and grad_alias.dtype == go2.dtype:
                ga = grad_alias.reshape(-1, xc.shape[-1])
                ga = ga if ga.is_contiguous() else ga.contiguous()
            own = ga is not None and UF.grad_tag(grad_alias, '_ubv_owned') and ga.data_ptr() == grad_alias.data_ptr()
            gx = UF.gemm_nt(go2, _transposed16(w), residual=ga, out=ga if own else None)
            if gx is not None:
                gx = gx.view(xc.shape)
                if grad_alias is not None and ga is None:
                    gx = gx + grad_alias.to(x_dtype)
        if gx is None and ctx.needs_input_grad[0]:
            if grad_alias is not None and grad_alias.dtype == go2.dtype == x_dtype:
                ga = grad_alias.reshape(-1, xc.shape[-1])
                if UF.grad_tag(grad_alias, '_ubv_owned') and ga.is_contiguous():
                    # a fresh tensor produced for this edge alone (functional._AddDropoutNorm marks
                    # its grad_identity): accumulate in place — out-of-place addmm would first copy
                    # it into the result.  Untagged gradients may be shared (AddBackward hands ONE
                    # tensor to both of its inputs) and are left untouched.
                    gx = ga.addmm_(go2, w).view(xc.shape)
                else:
                    gx = torch.addmm(ga, go2, w).view(xc.shape)
            else:
                gx = (go2 @ w).view(xc.shape).to(x_dtype)
                if grad_alias is not None:
                    gx = gx + grad_alias.to(x_dtype)
        x2 = xc.reshape(-1, xc.shape[-1])
        rows = x2.shape[0]
        gw = gb = None
        need_w = any(ctx.needs_input_grad[7:7 + n])
        need_b = has_bias and any(ctx.needs_input_grad[7 + n:])
        part = None
        if (need_w or need_b) and go2.is_cuda and _MFMA_WGRAD and go2.dtype == x2.dtype and \
                go2.is_contiguous() and x2.is_contiguous():
            # one pass over grad_out and x on the matrix cores: dW and the bias sums together
            from . import functional as UF
            res = UF.gemm_wgrad(go2, x2)
            if res is not None:
                gw, gb = res
                need_w = need_b_left = False
                grads = []
                off = 0
                for i in range(n):
                    grads.append(gw[off:off + outs[i]].to(pdt[i]) if ctx.needs_input_grad[7 + i] else None)
                    off += outs[i]
                off = 0
                for i in range(n if has_bias else 0):
                    grads.append(gb[off:off + outs[i]].to(pdt[n + i]))
                    off += outs[i]
                return (gx, None, None, None, None, None, grb, *grads)
        if need_w:
            s = _splits(rows)
            if s > 1:
                part = torch.bmm(go2.view(s, rows // s, -1).transpose(1, 2), x2.view(s, rows // s, -1))
            else:
                gw = (go2.t() @ x2).float()
        vec = 16 // go2.element_size()
        colsum = UF.grad_tag(grad_out, '_ubv_colsum')
        if need_b and colsum is not None and colsum.numel() == go2.shape[1]:
            gb, need_b = colsum, False          # summed by the kernel that produced grad_out
        if go2.is_cuda and (need_b or part is not None) and go2.shape[1] % vec == 0 and \
                go2.shape[1] // vec <= 256 and \
                (part is None or (part.dtype == go2.dtype and part[0].numel() % vec == 0)):
            # one launch: bias column sums + the sum over the split-K slices (library reductions)
            from . import functional as UF
            gb_k, gw_p = UF.linear_grad_reduce(go2 if need_b else None, part)
            gb = gb_k if need_b else gb
            gw = gw_p if part is not None else gw
        else:
            if part is not None:
                gw = part.sum(0, dtype=torch.float32)
            if need_b:
                gb = go2.sum(0, dtype=torch.float32)
        grads = []
        off = 0
        for i in range(n):
            grads.append(None if gw is None else gw[off:off + outs[i]].to(pdt[i]))
            off += outs[i]
        off = 0
        for i in range(n if has_bias else 0):
            grads.append(None if gb is None else gb[off:off + outs[i]].to(pdt[n + i]))
            off += outs[i]
        return (gx, None, None, None, None, None, grb, *grads)


_SELF_IN_SPLIT_FWD = os.environ.get('UBV_SELF_IN_SPLIT_FWD', '0') != '0'


class _SelfAttnIn(Function):
    """The three Linears on the query of a BEV self-attention as ONE GEMM: ``value = x . Wv^T + bv`` and ``offsets |
    logits = x . [Wo; Wa]^T + [bo; ba] + row_bias[q]`` (``ubv_gemm_nt_dual``: x is read once, the two results leave as two
    tensors), plus the pass-through alias of ``x`` for the residual branch.  Backward: ONE input-gradient GEMM over
    [d value | d offsets,logits] with the alias' gradient in its epilogue (in place when that tensor was made for this
    edge), ONE weight-gradient pass over x (``ubv_gemm_wgrad_dual``), and d(row_bias) = the batch sum of d(offsets |
    logits).  f32 CUDA tensors only; ``self_attn_in`` falls back to the separate Linears otherwise."""

    @staticmethod
    def forward(ctx, x, row_bias, wv, bv, wo, bo, wa, ba):
        from . import functional as UF
        split = _split_weights((wv, wo, wa))
        # [bv | bo | ba] from the per-step shadow buffers (refreshed ahead of the pass by one multi-tensor copy): a
        # torch.cat here is a framework kernel inside the encoders' two-stream window
        bias = _cached_lowp((bv, bo, ba), torch.float32) if (_ACTIVE and all(isinstance(b, torch.nn.Parameter) for b in (bv, bo, ba))) \
            else torch.cat((bv.detach(), bo.detach(), ba.detach())).float()
        n2 = wo.shape[0] + wa.shape[0]
        res = None
        if _SELF_IN_SPLIT_FWD and wv.shape[0] % 32 == 0 and n2 % 32 == 0 and x.shape[-1] in (64, 128, 192, 256) and \
                row_bias.stride(1) == 1 and row_bias.stride(0) % 4 == 0 and row_bias.data_ptr() % 16 == 0:
            # two launches of the weight-stationary GEMM — value (N = 256), then offsets | logits (N = 96) with the
            # positional term as its row-periodic residual — instead of ONE launch of the tile-per-block dual-output
            # kernel: x is read twice (the second time from the Infinity Cache), the kernel is the faster one
            # (UBV_SELF_IN_SPLIT_FWD=0: the dual kernel; A/B in profiles/r06_self_in_split.txt)
            n1 = wv.shape[0]
            v = UF.gemm_nt(x, split[0][:n1], split[1][:n1], bias=bias[:n1])
            ol = UF.gemm_nt(x, split[0][n1:], split[1][n1:], bias=bias[n1:], row_bias=row_bias) if v is not None else None
            if v is not None and ol is not None:
                res = (v, ol)
        if res is None:
            res = UF.gemm_nt_dual(x, split[0], split[1], bias=bias, y2_cols=n2, row_bias=row_bias)
        if res is None:
            raise RuntimeError('self_attn_in: shape outside ubv_gemm_nt_dual (checked by self_attn_in_supported)')
        ctx.save_for_backward(x, split[2], split[3])
        ctx.rows = row_bias.shape[0]
        ctx.rb_slot = getattr(row_bias, '_ubv_grad_slot', None)
        ctx.outs = (wv.shape[0], wo.shape[0], wa.shape[0])
        return res[0], res[1], x.view_as(x)

    @staticmethod
    def backward(ctx, gv, gol, galias):
        from . import functional as UF
        x, wth, wtl = ctx.saved_tensors
        C = x.shape[-1]
        gv2 = gv.reshape(-1, gv.shape[-1])
        gol2 = gol.reshape(-1, gol.shape[-1])
        gv2 = gv2 if gv2.is_contiguous() else gv2.contiguous()
        gol2 = gol2 if gol2.is_contiguous() else gol2.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            ga = None
            if galias is not None:
                ga = galias.reshape(-1, C)
                ga = ga if ga.is_contiguous() else ga.contiguous()
            own = ga is not None and UF.grad_tag(galias, '_ubv_owned') and ga.data_ptr() == galias.data_ptr()
            gx = UF.gemm_nt_dual(gv2, wth, wtl, x2=gol2, residual=ga, out=ga if own else None)
            if gx is None:
                raise RuntimeError('self_attn_in: input gradient outside ubv_gemm_nt_dual')
            gx = gx.view(x.shape)
        grb = None
        if ctx.needs_input_grad[1]:
            # (this library's reductions only: the framework's f32 add kernel holds packed f32 FMAs, and this runs inside an
            #  encoder branch, beside the other stream's MFMA kernels — DESIGN section 5 "Two streams")
            grb = _row_bias_grad(gol2.view(-1, ctx.rows, gol2.shape[-1]), ctx.rb_slot)
        # weight gradients: two passes over x measured FASTER than the fused ubv_gemm_wgrad_dual (90 vs 113 us back to
        # back at M = 80 000: the kernel is bound by its tiles' MFMA / LDS work, which is the same either way, and
        # the 352-row product takes 85 slabs of 6 tiles where the two take 128 x 4 and 256 x 2); UBV_SELF_IN_WGRAD=dual
        # keeps the fused form for A/B runs
        x2d = x.reshape(-1, C)
        if _SELF_IN_WGRAD_DUAL:
            res = UF.gemm_wgrad_dual(gv2, gol2, x2d)
            if res is None:
                raise RuntimeError('self_attn_in: weight gradient outside ubv_gemm_wgrad_dual')
            gw, gb = res
        else:
            r1, r2 = UF.gemm_wgrad(gv2, x2d), UF.gemm_wgrad(gol2, x2d)
            if r1 is None or r2 is None:
                raise RuntimeError('self_attn_in: weight gradient outside ubv_gemm_wgrad')
            gw = gb = None
        nv, no, na = ctx.outs
        if gw is not None:
            r1, r2 = (gw[:nv], gb[:nv]), (gw[nv:], gb[nv:])
        need = ctx.needs_input_grad
        return (gx, grb,
                r1[0] if need[2] else None, r1[1] if need[3] else None,
                r2[0][:no] if need[4] else None, r2[1][:no] if need[5] else None,
                r2[0][no:] if need[6] else None, r2[1][no:] if need[7] else None)


def self_attn_in_supported(x, row_bias, wv, wo, wa):
    """The fused GEMM takes f32 CUDA tensors, a contiguous query, a value width that is a multiple of 128 (its column
    tiles) and the split-bf16 MFMA path switched on."""
    ws = (wv, wo, wa)
    return _MFMA_F32 and _FUSE_SELF_IN and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and \
        not torch.is_autocast_enabled('cuda') and row_bias is not None and row_bias.dtype == torch.float32 and \
        all(w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == x.shape[-1] for w in ws) and \
        x.shape[-1] % 32 == 0 and wv.shape[0] % 128 == 0 and (wo.shape[0] + wa.shape[0]) % 32 == 0


_FUSE_SELF_IN = os.environ.get('UBV_FUSE_SELF_IN', '1') != '0'
_SELF_IN_WGRAD_DUAL = os.environ.get('UBV_SELF_IN_WGRAD', '') == 'dual'


def self_attn_in(x, row_bias, wv, bv, wo, bo, wa, ba):
    """(value, offsets | logits, alias of x) — see ``_SelfAttnIn``; check ``self_attn_in_supported`` first."""
    return _SelfAttnIn.apply(x, row_bias, wv, bv, wo, bo, wa, ba)


def _run(x, weights, biases, passthru=False, act=None, row_bias=None):
    has_bias = biases[0] is not None
    params = list(weights) + (list(biases) if has_bias else [])
    if x.is_cuda and torch.is_autocast_enabled('cuda'):
        # no autocast(enabled=False) scope here: every operand of the GEMM inside is already in the
        # autocast dtype, so the ambient policy has nothing to cast (and the context manager costs
        # ~1 us x 3 per call on a host-bound forward)
        return _Linear.apply(x, torch.get_autocast_dtype('cuda'), len(weights), has_bias, passthru, act,
                             row_bias, *params)
    return _Linear.apply(x, x.dtype if x.dtype == weights[0].dtype else weights[0].dtype,
                         len(weights), has_bias, passthru, act, row_bias, *params)


def linear(x, weight, bias=None):
    """``F.linear(x, weight, bias)``; follows the ambient autocast dtype."""
    return _run(x, [weight], [bias])


def linear_relu_dropout(x, weight, bias=None, p=0.0, training=False, passthru=False):
    """``dropout(relu(F.linear(x, weight, bias)), p)`` with the activation in the GEMM epilogue (the first
    half of an FFN).  Feed the result ONLY to ``linear_after_relu_dropout`` with the same ``p`` /
    ``training``: that Linear's input gradient comes back already multiplied by the activation's
    derivative.  ``passthru``: as ``linear_pass``."""
    return _run(x, [weight], [bias], passthru=passthru, act=('relu_drop', float(p) if training else 0.0))


def linear_after_relu_dropout(a, weight, bias=None, p=0.0, training=False):
    """``F.linear(a, weight, bias)`` for ``a = linear_relu_dropout(...)`` (the second half of an FFN).  Marks the
    producer's autograd node: its backward then insists on receiving exactly this Linear's (tagged) input gradient."""
    fn = getattr(a, 'grad_fn', None)
    if fn is not None:
        try:
            fn.ubv_folded = True
        except AttributeError:
            pass
    return _run(a, [weight], [bias], act=('masked_in', float(p) if training else 0.0))


def linear_cat(x, weights, biases):
    """``F.linear(x, cat(weights), cat(biases))`` — several Linear layers on one input as one GEMM."""
    return _run(x, list(weights), list(biases))


def linear_pass(x, weight, bias=None):
    """``(F.linear(x, weight, bias), x')`` with ``x'`` an alias of ``x`` for the residual branch
    (see ``_Linear``): use ``x'`` wherever ``x`` would be used again."""
    return _run(x, [weight], [bias], passthru=True)


def linear_cat_pass(x, weights, biases, row_bias=None):
    """``linear_cat`` with the pass-through alias of ``linear_pass``; ``row_bias`` [R, N]: + row_bias[m % R] on output
    row m (a term shared by the samples of the batch, see ``_Linear``)."""
    return _run(x, list(weights), list(biases), passthru=True, row_bias=row_bias)
