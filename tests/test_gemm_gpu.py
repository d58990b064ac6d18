"""The hand-written MFMA GEMM (ubv_gemm_nt) against f64 products: split-bf16 arithmetic on f32 data
(x_hi w_hi + x_hi w_lo + x_lo w_hi, ~2^-17 per product), single product on 16-bit data; bias, residual
(aliasing the output), ragged row counts, every column tiling."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('M,N,K', [(1000, 256, 256), (4099, 192, 256), (130, 96, 512), (777, 512, 256),
                                   (64, 32, 64), (2112, 128, 320)])
def test_gemm_nt_split_f32(M, N, K):
    from unibev_amd.functional import gemm_nt, split_weight
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    wh, wl, wth, wtl = split_weight(w.to(DEV))
    # the halves reassemble the weight to 2^-17, in both orientations
    torch.testing.assert_close(wh.float() + wl.float(), w.to(DEV), rtol=2.0 ** -16, atol=1e-7)
    assert torch.equal(wth, wh.t().contiguous()) and torch.equal(wtl, wl.t().contiguous())
    ref = x.double() @ w.double().t()
    scale = float(ref.abs().max())
    y = gemm_nt(x.to(DEV), wh, wl)
    assert y is not None and y.dtype == torch.float32
    assert float((y.cpu().double() - ref).abs().max()) < 3e-5 * scale
    y = gemm_nt(x.to(DEV), wh, wl, bias=b.to(DEV), residual=r.to(DEV))
    assert float((y.cpu().double() - (ref + b.double() + r.double())).abs().max()) < 3e-5 * scale
    # residual aliasing the output (the in-place accumulation of the input-gradient GEMM)
    acc = r.to(DEV).clone()
    y = gemm_nt(x.to(DEV), wh, wl, residual=acc, out=acc)
    assert y.data_ptr() == acc.data_ptr()
    assert float((acc.cpu().double() - (ref + r.double())).abs().max()) < 3e-5 * scale
    # input gradient: dX = dY . W through the transposed halves
    gy = torch.randn(M, N, generator=g)
    gx = gemm_nt(gy.to(DEV), wth, wtl)
    assert gx is not None
    if True:
        refx = gy.double() @ w.double()
        assert float((gx.cpu().double() - refx).abs().max()) < 3e-5 * float(refx.abs().max())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3001, 192, 256), (286, 64, 288), (1000, 128, 96), (500, 256, 2304)])
def test_gemm_nt_16bit(dtype, shape):
    # (K % 64 != 0 walks 32-deep chunks: the staged epilogue then needs more LDS than the operand tiles)
    from unibev_amd.functional import gemm_nt
    g = torch.Generator(device='cpu').manual_seed(3)
    M, N, K = shape
    x = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g)
    y = gemm_nt(x.to(DEV), w.to(DEV), bias=b.to(DEV))
    assert y.dtype == dtype
    ref = x.double() @ w.double().t() + b.double()
    tol = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)          # one rounding of the output
    assert float((y.cpu().double() - ref).abs().max()) < tol * float(ref.abs().max())


def test_gemm_nt_declines_shapes_it_cannot_tile():
    from unibev_amd.functional import gemm_nt, split_weight
    wh, wl, _, _ = split_weight(torch.randn(24, 64, device=DEV), transposed=False)
    assert gemm_nt(torch.randn(10, 64, device=DEV), wh, wl) is None          # N % 32 != 0
    wh, wl, _, _ = split_weight(torch.randn(32, 48, device=DEV), transposed=False)
    assert gemm_nt(torch.randn(10, 48, device=DEV), wh, wl) is None          # K % 32 != 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,N,K', [(5000, 256, 256), (4099, 192, 256), (777, 96, 128), (9000, 512, 256),
                                   (300, 64, 40), (12345, 256, 512)])
def test_gemm_wgrad(M, N, K, dtype):
    if K == 40 and dtype == torch.float32:
        K = 36                                              # f32 rows load 4 columns at a time, 16-bit rows 8
    """dW = dY^T X and db = column sums of dY in one pass (ubv_gemm_wgrad + the slab sum) against f64:
    split-bf16 products for f32 data (~2^-17 each), exact products of the stored values for 16-bit data;
    ragged row counts, partial tiles in N and K."""
    from unibev_amd.functional import gemm_wgrad
    g = torch.Generator(device='cpu').manual_seed(M + N)
    gy = torch.randn(M, N, generator=g).to(dtype)
    x = torch.randn(M, K, generator=g).to(dtype)
    res = gemm_wgrad(gy.to(DEV), x.to(DEV))
    assert res is not None
    gw, gb = res
    assert gw.shape == (N, K) and gb.shape == (N,) and gw.dtype == torch.float32
    refw = gy.double().t() @ x.double()
    refb = gy.double().sum(0)
    tol = 3e-5 if dtype == torch.float32 else 2e-6           # 16-bit inputs multiply exactly; f32 accumulation
    assert float((gw.cpu().double() - refw).abs().max()) < max(tol * float(refw.abs().max()), 1e-3 * tol * M ** 0.5 * 30)
    assert float((gb.cpu().double() - refb).abs().max()) < 1e-5 * max(1.0, float(refb.abs().max())) * (1 if dtype == torch.float32 else 1)


@pytest.fixture
def wgrad_kernel():
    """Selects the weight-gradient kernel for one test (ubv_debug_set_wgrad_ws) and restores the default."""
    from unibev_amd._lib import lib
    def choose(pw):
        assert lib().ubv_debug_set_wgrad_ws(pw, -1) == 0
    yield choose
    lib().ubv_debug_set_wgrad_ws(0, -1)


@pytest.mark.parametrize('pw', [4, 8])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,N,K', [(5000, 256, 256), (4099, 192, 256), (777, 96, 128), (9000, 512, 256), (300, 64, 40),
                                   (12345, 256, 512), (64, 256, 256), (129, 128, 128), (40000, 256, 256)])
def test_gemm_wgrad_wave_specialised_kernel(M, N, K, dtype, pw, wgrad_kernel):
    """csrc/gemm_wgrad_ws.inl (4 MFMA waves + 4 or 8 producer waves, one block per CU, double-buffered operand planes)
    against f64 with the 4-wave kernel's tolerances, and against the 4-wave kernel itself: same products, same order
    inside a slab; the slabs differ (half as many), so the comparison is to f32 summation noise, not to the bit.
    Ragged row counts (one chunk, a chunk + 1 row, slabs with an odd chunk count), partial tiles in N and K."""
    from unibev_amd.functional import gemm_wgrad
    if K == 40 and dtype == torch.float32:
        K = 36
    g = torch.Generator(device='cpu').manual_seed(M + N + pw)
    gy = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    base_w, base_b = gemm_wgrad(gy, x)
    wgrad_kernel(pw)
    res = gemm_wgrad(gy, x)
    assert res is not None
    gw, gb = res
    refw = gy.double().t() @ x.double()
    refb = gy.double().sum(0)
    tol = 3e-5 if dtype == torch.float32 else 2e-6
    assert float((gw.double() - refw).abs().max()) < max(tol * float(refw.abs().max()), 1e-3 * tol * M ** 0.5 * 30)
    assert float((gb.double() - refb).abs().max()) < 1e-5 * max(1.0, float(refb.abs().max()))
    assert float((gw - base_w).abs().max()) <= 4e-6 * float(refw.abs().max()) + 1e-6 * M ** 0.5
    assert float((gb - base_b).abs().max()) <= 4e-6 * max(1.0, float(refb.abs().max()))


@pytest.mark.parametrize('pw', [4, 8])
def test_gemm_wgrad_dual_wave_specialised_kernel(pw, wgrad_kernel):
    """ubv_gemm_wgrad_dual (grad_out in two matrices, the fused value_proj | offsets | logits Linear) on the
    wave-specialised kernel against the product of the concatenated matrix in f64."""
    from unibev_amd.functional import gemm_wgrad_dual
    g = torch.Generator(device='cpu').manual_seed(11)
    M, N1, N2, K = 7001, 256, 96, 256
    g1 = torch.randn(M, N1, generator=g).to(DEV)
    g2 = torch.randn(M, N2, generator=g).to(DEV)
    x = torch.randn(M, K, generator=g).to(DEV)
    wgrad_kernel(pw)
    res = gemm_wgrad_dual(g1, g2, x)
    assert res is not None
    gw, gb = res
    gy = torch.cat([g1, g2], 1).double()
    refw, refb = gy.t() @ x.double(), gy.sum(0)
    assert gw.shape == (N1 + N2, K)
    assert float((gw.double() - refw).abs().max()) < 3e-5 * float(refw.abs().max())
    assert float((gb.double() - refb).abs().max()) < 1e-5 * float(refb.abs().max())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p', [0.0, 0.1])
def test_ffn_activation_in_gemm_epilogues(dtype, p, monkeypatch):
    """FFN (Linear -> ReLU -> Dropout -> Linear): the activation in the first GEMM's epilogue and its
    derivative in the second GEMM's input-gradient epilogue (ubv_gemm_nt_act) against the same module with
    the separate relu_dropout kernels (UBV_FFN_FUSE=0) — same seeds, so the same keep mask — and, without
    dropout, against plain torch in f64."""
    import contextlib
    from unibev_amd.registry import build_feedforward_network
    from unibev_amd import functional as UF
    torch.manual_seed(5)
    ffn = build_feedforward_network(dict(type='FFN', embed_dims=64, feedforward_channels=128, ffn_drop=p,
                                         add_identity=True)).to(DEV).train()
    x0 = torch.randn(2, 700, 64, device=DEV)
    gy = torch.randn(2, 700, 64, device=DEV)
    ctx = contextlib.nullcontext() if dtype == torch.float32 else torch.autocast('cuda', dtype=dtype)

    def run(fuse):
        monkeypatch.setenv('UBV_FFN_FUSE', '1' if fuse else '0')
        torch.manual_seed(11)
        UF.new_step()
        for q in ffn.parameters():
            q.grad = None
        x = x0.clone().requires_grad_()
        with ctx:
            parts = ffn.forward_parts(x)
            assert parts is not None
            out = parts[0]
        out.float().backward(gy)
        return [out.float().detach(), x.grad.clone()] + [q.grad.clone() for q in ffn.parameters()]

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        if dtype == torch.float32:
            torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-5)
        else:       # the fused path rounds the hidden activation once instead of twice: compare in norm
            assert float((u - v).norm() / v.norm().clamp_min(1e-6)) < 2e-2
    if p == 0.0 and dtype == torch.float32:
        lin1, lin2 = ffn.layers[0][0], ffn.layers[1]
        x = x0.double().clone().requires_grad_()
        w1, b1, w2, b2 = [t.detach().double().requires_grad_() for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
        ref = torch.relu(x @ w1.t() + b1) @ w2.t() + b2
        ref.backward(gy.double())
        torch.testing.assert_close(a[0].double(), ref.detach(), rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(a[1].double(), x.grad, rtol=2e-4, atol=2e-4)
        for got, want in zip(a[2:], (w1.grad, b1.grad, w2.grad, b2.grad)):
            torch.testing.assert_close(got.double(), want, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize('max_norm', [None, 0.5])
def test_flat_adamw_matches_torch_adamw_with_clipping(max_norm):
    """ubv_sumsq_f32 + ubv_adamw_flat over flat buffers against torch.nn.utils.clip_grad_norm_ +
    torch.optim.AdamW, five steps, odd tensor sizes (tail handling)."""
    from unibev_amd.dp import FlatGradients
    from unibev_amd.optim import FlatAdamW
    torch.manual_seed(2)
    shapes = [(257, 33), (5,), (64, 64), (3, 7, 11), (1,)]
    mine = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    fg = FlatGradients(mine)
    fg.attach()
    opt = FlatAdamW(mine, fg, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_grad_norm=max_norm)
    topt = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    for it in range(5):
        gs = [torch.randn(*s, device=DEV) * (0.1 + it) for s in shapes]
        for p, r, g in zip(mine, ref, gs):
            p.grad.copy_(g)
            r.grad = g.clone()
        if max_norm is not None:
            tn = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        topt.step()
        opt.step()
        if max_norm is not None:
            assert abs(opt.grad_norm() - float(tn)) < 1e-4 * float(tn)
        for p, r in zip(mine, ref):
            torch.testing.assert_close(p.detach(), r.detach(), rtol=2e-6, atol=2e-7)
    assert all(p.data_ptr() >= opt.flat.data_ptr() for p in mine)      # parameters live in the flat buffer


def test_flat_adamw_parameter_groups_match_torch_param_groups():
    """ubv_adamw_flat_groups: runs of parameters with their own lr / weight decay (the reference's paramwise_cfg,
    config :455-462) against torch.optim.AdamW with the same param_groups; range boundaries fall inside a 16-byte
    load (odd sizes), one clip coefficient over all groups."""
    from unibev_amd.dp import FlatGradients
    from unibev_amd.optim import FlatAdamW, paramwise_groups
    torch.manual_seed(4)
    names = ['img_backbone.conv.weight', 'img_backbone.conv.bias', 'head.fc.weight', 'head.fc.bias', 'img_backbone.tail']
    shapes = [(37, 9), (5,), (64, 33), (3,), (1,)]
    mine = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    cfg = dict(custom_keys={'img_backbone': dict(lr_mult=0.1), 'fc.bias': dict(decay_mult=0.0)})
    fg = FlatGradients(mine)
    fg.attach()
    opt = FlatAdamW(paramwise_groups(zip(names, mine), 3e-3, 0.05, **cfg), fg, lr=3e-3, weight_decay=0.05,
                    max_grad_norm=0.5)
    assert len(opt.ranges) == 4
    topt = torch.optim.AdamW(paramwise_groups(zip(names, ref), 3e-3, 0.05, **cfg), lr=3e-3, weight_decay=0.05)
    for it in range(5):
        gs = [torch.randn(*s, device=DEV) * (0.1 + it) for s in shapes]
        for p, r, g in zip(mine, ref, gs):
            p.grad.copy_(g)
            r.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        topt.step()
        opt.step()
        for p, r in zip(mine, ref):
            torch.testing.assert_close(p.detach(), r.detach(), rtol=2e-6, atol=2e-7)


def test_flat_adamw_follows_a_learning_rate_schedule():
    """A scheduler changes lr every iteration (the reference: linear warm-up + CosineAnnealing, config :463-469).  Both
    ways of driving it — mmcv's LrUpdaterHook writing ``group['lr']`` from ``group['initial_lr']``, and ``opt.lr = x`` —
    must reach the kernel in the NEXT step (the per-range arrays were filled once at construction)."""
    import math
    from unibev_amd.dp import FlatGradients
    from unibev_amd.optim import FlatAdamW, paramwise_groups
    torch.manual_seed(6)
    names = ['img_backbone.w', 'head.w', 'head.b']
    shapes = [(19, 7), (33, 5), (11,)]
    mine = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    cfg = dict(custom_keys={'img_backbone': dict(lr_mult=0.1)})
    fg = FlatGradients(mine)
    fg.attach()
    opt = FlatAdamW(paramwise_groups(zip(names, mine), 2e-3, 0.05, **cfg), fg, lr=2e-3, weight_decay=0.05)
    topt = torch.optim.AdamW(paramwise_groups(zip(names, ref), 2e-3, 0.05, **cfg), lr=2e-3, weight_decay=0.05)
    for g in topt.param_groups:
        g.setdefault('initial_lr', g['lr'])
    assert [g['initial_lr'] for g in opt.param_groups] == [2e-4, 2e-3]
    for it in range(8):
        factor = (it + 1) / 4 if it < 4 else 0.5 * (1 + math.cos(math.pi * (it - 4) / 4))     # warm-up, then cosine
        for g in topt.param_groups:
            g['lr'] = g['initial_lr'] * factor
        if it % 2:
            for g in opt.param_groups:
                g['lr'] = g['initial_lr'] * factor
        else:
            opt.lr = 2e-3 * factor
            if it == 4:
                opt.weight_decay = 0.02
        if it >= 4:
            for g in topt.param_groups:
                g['weight_decay'] = 0.02
        assert [g['lr'] for g in opt.param_groups] == pytest.approx([2e-4 * factor, 2e-3 * factor])
        gs = [torch.randn(*s, device=DEV) for s in shapes]
        for p, r, g in zip(mine, ref, gs):
            p.grad.copy_(g)
            r.grad = g.clone()
        topt.step()
        opt.step()
        for p, r in zip(mine, ref):
            torch.testing.assert_close(p.detach(), r.detach(), rtol=2e-6, atol=2e-7)


def test_split_weights_batched_matches_the_per_weight_kernel():
    from unibev_amd import functional as UF
    g = torch.Generator(device='cpu').manual_seed(5)
    groups = [[torch.randn(256, 256, generator=g).to(DEV)],
              [torch.randn(64, 256, generator=g).to(DEV), torch.randn(32, 256, generator=g).to(DEV)],      # concatenated
              [torch.randn(512, 256, generator=g).to(DEV)], [torch.randn(100, 72, generator=g).to(DEV)]]
    groups += [[torch.randn(40, 33, generator=g).to(DEV)] for _ in range(60)]                               # > one launch
    got = UF.split_weights_batched(groups)
    for grp, out in zip(groups, got):
        want = UF.split_weight(torch.cat(grp, 0))
        for a, b in zip(out, want):
            assert a.shape == b.shape and torch.equal(a, b)


def test_second_pass_takes_the_presplit_weights_and_follows_weight_updates():
    import unibev_amd.linear as UL
    lin = torch.nn.Linear(256, 96).to(DEV)
    x = torch.randn(300, 256, device=DEV)
    UL.clear_lowp_cache()
    outs = []
    for it in range(3):
        with UL.lowp_step_cache():
            if it > 0:
                assert len(UL._SPLIT_PASS) == 1          # split on entry, before the layer asked for it
            outs.append(UL.linear(x, lin.weight, lin.bias))
        with torch.no_grad():
            lin.weight.mul_(2.0)
    assert torch.allclose(outs[1], 2 * outs[0] - lin.bias.detach(), rtol=1e-4, atol=1e-4)
    ref = torch.nn.functional.linear(x, lin.weight / 2, lin.bias)
    assert float((outs[2] - ref).detach().abs().max()) < 1e-3 * float(ref.detach().abs().max())


def test_fused_ffn_activation_refuses_a_second_consumer():
    """``linear_relu_dropout`` -> ``linear_after_relu_dropout`` folds the activation's derivative into the second
    Linear's input gradient.  If the activation also feeds something else, autograd sums a folded and a raw gradient:
    the backward must fail loudly instead of applying the derivative to the sum."""
    from unibev_amd.linear import linear_after_relu_dropout, linear_relu_dropout
    torch.manual_seed(0)
    x = torch.randn(300, 64, device=DEV, requires_grad=True)
    w1 = torch.randn(128, 64, device=DEV, requires_grad=True)
    w2 = torch.randn(64, 128, device=DEV, requires_grad=True)
    a = linear_relu_dropout(x, w1, None, 0.1, True)
    y = linear_after_relu_dropout(a, w2, None, 0.1, True)
    (y.sum() + a.sum()).backward.__self__       # (build the graph with two consumers of `a`)
    with pytest.raises(RuntimeError, match='besides'):
        (y.sum() + a.sum()).backward()
    # the sanctioned use works, and so does the activation without a folding consumer
    a = linear_relu_dropout(x, w1, None, 0.0, True)
    linear_after_relu_dropout(a, w2, None, 0.0, True).sum().backward()
    a = linear_relu_dropout(x.detach().requires_grad_(), w1, None, 0.0, True)
    a.sum().backward()


@pytest.mark.parametrize('view', [False, True])
def test_gemm_nt_row_periodic_bias_and_linear_row_bias(view):
    """ubv_gemm_nt_rowbias: + row_bias[m % R] in the epilogue (the positional term of the BEV self-attention's
    offset / logit Linears), through the operator and through linear_cat_pass with its gradients."""
    from unibev_amd import functional as UF
    from unibev_amd.linear import linear_cat_pass
    torch.manual_seed(3)
    B, R, K, N = 3, 520, 256, 96
    x = torch.randn(B, R, K, device='cuda')
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    b = torch.randn(N, device='cuda')
    table = torch.randn(R, 3 * N, device='cuda')
    rb = table[:, N:2 * N] if view else table[:, :N].contiguous()
    wh, wl, _, _ = UF.split_weight(w)
    y = UF.gemm_nt(x, wh, wl, bias=b, row_bias=rb)
    ref = (x.double() @ w.double().t() + b.double() + rb.double()[None]).float()
    assert y is not None
    torch.testing.assert_close(y, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    # autograd: two weights sharing the input, pass-through alias, row_bias gradient = sum over the batch
    w1 = torch.nn.Parameter(w[:64].clone()); w2 = torch.nn.Parameter(w[64:].clone())
    b1 = torch.nn.Parameter(b[:64].clone()); b2 = torch.nn.Parameter(b[64:].clone())
    xg = x.clone().requires_grad_()
    rbg = rb.detach().clone().requires_grad_()
    out, alias = linear_cat_pass(xg, (w1, w2), (b1, b2), row_bias=rbg)
    cot, cot2 = torch.randn_like(out), torch.randn_like(xg)
    ((out * cot).sum() + (alias * cot2).sum()).backward()
    xr = x.double().requires_grad_()
    wr = w.double().requires_grad_(); br = b.double().requires_grad_(); rr = rb.double().detach().clone().requires_grad_()
    outr = xr @ wr.t() + br + rr[None]
    ((outr * cot.double()).sum() + (xr * cot2.double()).sum()).backward()
    torch.testing.assert_close(out.detach(), outr.detach().float(), rtol=2e-5, atol=2e-5 * float(outr.detach().abs().max()))
    tol = lambda t: dict(rtol=2e-4, atol=2e-4 * float(t.abs().max()))       # noqa: E731
    torch.testing.assert_close(xg.grad, xr.grad.float(), **tol(xr.grad))
    torch.testing.assert_close(rbg.grad, rr.grad.float(), **tol(rr.grad))
    torch.testing.assert_close(torch.cat((w1.grad, w2.grad)), wr.grad.float(), **tol(wr.grad))
    torch.testing.assert_close(torch.cat((b1.grad, b2.grad)), br.grad.float(), **tol(br.grad))


def test_self_attn_in_fused_gemm_matches_separate_linears():
    """ubv_gemm_nt_dual / ubv_gemm_wgrad_dual through linear.self_attn_in: value_proj | offsets | logits in one GEMM with
    two outputs and the row-periodic positional term on the second, one input-gradient GEMM over both output gradients
    with the residual's gradient in its epilogue, one weight-gradient pass — against fp64 torch."""
    from unibev_amd.linear import self_attn_in, self_attn_in_supported
    torch.manual_seed(5)
    B, R, C, NV, NO, NA = 2, 700, 256, 256, 64, 32
    x = torch.randn(B, R, C, device='cuda')
    mk = lambda n: torch.nn.Parameter(torch.randn(n, C, device='cuda') / C ** 0.5)      # noqa: E731
    wv, wo, wa = mk(NV), mk(NO), mk(NA)
    bv, bo, ba = (torch.nn.Parameter(torch.randn(n, device='cuda')) for n in (NV, NO, NA))
    table = torch.randn(R, 2 * (NO + NA), device='cuda')
    rb = table[:, NO + NA:].detach().requires_grad_()           # a view with a row stride, as the encoders pass it
    xg = x.clone().requires_grad_()
    assert self_attn_in_supported(xg, rb, wv, wo, wa)
    v, ol, alias = self_attn_in(xg, rb, wv, bv, wo, bo, wa, ba)
    cv, co, ca = torch.randn_like(v), torch.randn_like(ol), torch.randn_like(alias)
    ((v * cv).sum() + (ol * co).sum() + (alias * ca).sum()).backward()
    d = lambda t: t.detach().double().requires_grad_()          # noqa: E731
    xr, rr = d(x), d(rb)
    pr = [d(p) for p in (wv, bv, wo, bo, wa, ba)]
    vr = xr @ pr[0].t() + pr[1]
    olr = xr @ torch.cat((pr[2], pr[4])).t() + torch.cat((pr[3], pr[5])) + rr[None]
    ((vr * cv.double()).sum() + (olr * co.double()).sum() + (xr * ca.double()).sum()).backward()
    tol = lambda t, r=2e-5: dict(rtol=r, atol=r * float(t.detach().abs().max()))       # noqa: E731
    torch.testing.assert_close(v.detach(), vr.detach().float(), **tol(vr))
    torch.testing.assert_close(ol.detach(), olr.detach().float(), **tol(olr))
    torch.testing.assert_close(xg.grad, xr.grad.float(), **tol(xr.grad, 2e-4))
    torch.testing.assert_close(rb.grad, rr.grad.float(), **tol(rr.grad, 2e-4))
    for got, ref in zip((wv, bv, wo, bo, wa, ba), pr):
        torch.testing.assert_close(got.grad, ref.grad.float(), **tol(ref.grad, 2e-4))


@pytest.mark.parametrize('M', [1, 31, 32, 33, 4099, 20000])
@pytest.mark.parametrize('N,K', [(256, 256), (512, 256), (192, 256), (128, 256), (256, 192), (256, 128), (768, 96), (256, 64),
                                 (192, 192), (128, 64)])
def test_gemm_ws_weight_stationary_kernel(M, N, K):
    """The weight-stationary persistent kernel (csrc/gemm_ws.hip: every f32 Linear with K <= 256 and N in {128, 192, 256 k})
    against f64: plain, bias + residual (also in place), the row-periodic term (ubv_gemm_nt_rowbias) and the masked form
    (ubv_gemm_nt_act mode 2); row counts around the 32-row tile, fewer tiles than blocks, several column groups."""
    from unibev_amd.functional import gemm_nt, gemm_nt_act, split_weight
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    wh, wl, _, _ = split_weight(w.to(DEV), transposed=False)
    ref = x.double() @ w.double().t()
    scale = max(float(ref.abs().max()), 1.0)
    xd = x.to(DEV)
    y = gemm_nt(xd, wh, wl)
    assert float((y.cpu().double() - ref).abs().max()) < 3e-5 * scale
    y = gemm_nt(xd, wh, wl, bias=b.to(DEV), residual=r.to(DEV))
    assert float((y.cpu().double() - (ref + b.double() + r.double())).abs().max()) < 3e-5 * scale
    acc = r.to(DEV).clone()
    y = gemm_nt(xd, wh, wl, residual=acc, out=acc)
    assert y.data_ptr() == acc.data_ptr()
    assert float((acc.cpu().double() - (ref + r.double())).abs().max()) < 3e-5 * scale
    # row-periodic term: R[m % period] (period divides M), through a view with a wider row stride
    for period in {1, M} | ({M // 7} if M % 7 == 0 and M > 7 else set()):
        wide = torch.randn(period, N + 32, generator=g)
        rb = wide.to(DEV)[:, 16:16 + N]
        y = gemm_nt(xd, wh, wl, bias=b.to(DEV), row_bias=rb)
        want = ref + b.double() + wide[:, 16:16 + N].double().repeat(M // period, 1)
        assert float((y.cpu().double() - want).abs().max()) < 3e-5 * scale
    # masked gradient form: (x @ w^T) / (1 - p) where mask != 0
    mask = (torch.rand(M, N, generator=g) < 0.6).float() * torch.rand(M, N, generator=g).clamp_min(0.1)
    y = gemm_nt_act(xd, wh, wl, act=2, mask=mask.to(DEV), p=0.25)
    want = torch.where(mask != 0, ref / 0.75, torch.zeros_like(ref))
    assert float((y.cpu().double() - want).abs().max()) < 3e-5 * scale / 0.75
    # FFN activation in the epilogue: dropout(relu(x @ w^T + b)) with the keep mask of the separate kernel for the seed
    from unibev_amd.functional import relu_dropout_raw
    for p in (0.0, 0.3):
        y = gemm_nt_act(xd, wh, wl, bias=b.to(DEV), act=1, p=p, seed=1234 + M)
        want = relu_dropout_raw(gemm_nt(xd, wh, wl, bias=b.to(DEV)), p, 1234 + M)
        assert torch.equal(y, want)
        if p == 0.0:
            assert float((y.cpu().double() - torch.relu(ref + b.double())).abs().max()) < 3e-5 * scale
        elif M >= 4099:                                     # about p of the positive entries are dropped
            pos = (ref + b.double()) > 1e-3
            assert 0.27 < float((y.cpu()[pos] == 0).float().mean()) < 0.33


@pytest.mark.parametrize('form', ['plain', 'residual', 'masked', 'relu_dropout'])
def test_gemm_ws_results_do_not_depend_on_memory_latency(form):
    """The weight-stationary GEMM counts its vector-memory instructions by hand (`s_waitcnt vmcnt(N)` in front of the LDS
    reads of DMA-filled tiles).  A count that is too high would read a tile before it landed — only when the memory system
    is slow enough.  So: 30 launches at the step's size while a second stream saturates HBM with 1 GB copies, every output
    bitwise equal to the launch made alone."""
    from unibev_amd.functional import gemm_nt, gemm_nt_act, split_weight
    g = torch.Generator(device='cpu').manual_seed(9)
    M, N, K = 80000, 256, 256
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV)
    wh, wl, _, _ = split_weight(w, transposed=False)

    def run():
        if form == 'plain':
            return gemm_nt(x, wh, wl, bias=b)
        if form == 'residual':
            return gemm_nt(x, wh, wl, bias=b, residual=r)
        if form == 'masked':
            return gemm_nt_act(x, wh, wl, act=2, mask=r, p=0.1)
        return gemm_nt_act(x, wh, wl, bias=b, act=1, p=0.1, seed=77)
    ref = run().clone()
    torch.cuda.synchronize()
    big_a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=DEV)
    big_b = torch.empty_like(big_a)
    side = torch.cuda.Stream()
    outs = []
    for i in range(30):
        with torch.cuda.stream(side):
            big_b.copy_(big_a)
            big_a.copy_(big_b)
        outs.append(run())
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
