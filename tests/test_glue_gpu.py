"""Direct parity tests of the memory-bound glue kernels (flatten+embed, fusion epilogue) against the
oracle's restatement of transformer_fusion.py, forward and backward, including ragged sizes."""
import numpy as np
import pytest
import torch

from oracle import unibev_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('N,groups,C,HW', [(12, 6, 256, 176), (2, 1, 128, 33 * 31), (6, 3, 64, 5),
                                           (1, 1, 256, 180 * 180)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_flatten_embed_vs_oracle(N, groups, C, HW, dtype):
    """_pre_process_img_feats / _pre_process_pts_feats: (N,C,HW) -> (N,HW,C) + cams_embeds + level."""
    from unibev_amd.functional import flatten_embed
    torch.manual_seed(0)
    feat = torch.randn(N, C, HW, device=DEV).to(dtype)
    ea = torch.randn(groups, C, device=DEV)
    eb = torch.randn(C, device=DEV)
    cot = torch.randn(N, HW, C, device=DEV)
    f1, a1, b1 = feat.clone().requires_grad_(), ea.clone().requires_grad_(), eb.clone().requires_grad_()
    out = flatten_embed(f1, a1 if groups > 1 else None, b1)
    (out.float() * cot).sum().backward()
    # oracle: transformer_fusion.py:241-245 (bs = N/groups samples, `groups` cameras)
    f2, a2, b2 = (v.detach().float().cpu().requires_grad_() for v in (feat, ea, eb))
    P = {'cams_embeds': a2, 'img_level_embeds': b2[None], 'pts_level_embeds': b2[None]}
    if groups > 1:
        flat, _ = R.pre_process_img_feats(P, [f2.view(N // groups, groups, C, HW, 1)])
        ref = flat.permute(2, 0, 1, 3).reshape(N, HW, C)        # (Nc,hw,bs,C) -> (bs*Nc,hw,C)
    else:
        flat, _ = R.pre_process_pts_feats(P, [f2.view(N, C, HW, 1)])
        ref = flat.permute(1, 0, 2)
    (ref * cot.cpu()).sum().backward()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    torch.testing.assert_close(out.float().cpu(), ref.detach(), rtol=tol, atol=tol)
    torch.testing.assert_close(f1.grad.float().cpu(), f2.grad, rtol=tol, atol=tol)
    # embedding gradients are sums over N*HW rows of grad_out, which is rounded to `dtype`
    eps = 1e-6 if dtype == torch.float32 else 2 ** -8
    gtol = dict(rtol=1e-4, atol=10 * eps * (N * HW) ** 0.5 + 1e-4)
    torch.testing.assert_close(b1.grad.cpu(), b2.grad, **gtol)
    if groups > 1:
        torch.testing.assert_close(a1.grad.cpu(), a2.grad, **gtol)


@pytest.mark.parametrize('method,norm,flags', [('linear', 'ChannelNormWeights', (1, 1)),
                                               ('linear', 'ChannelNormWeights', (1, 0)),
                                               ('avg', None, (1, 1)), ('cat', None, (0, 1)),
                                               ('cat', None, (1, 1))])
@pytest.mark.parametrize('spatial', [False, True])
def test_bev_fuse_vs_oracle(method, norm, flags, spatial):
    """channel_feature_norm + spatial_feature_norm + multi_modal_fusion + permute, with modality
    flags, forward and backward to features and to the CNW / spatial parameters."""
    from unibev_amd.functional import bev_fuse
    torch.manual_seed(1)
    B, Nq, C = 2, 333, 128
    c_flag, l_flag = flags
    img = torch.randn(B, Nq, C, device=DEV)
    pts = torch.randn(B, Nq, C, device=DEV)
    cwp = {k: torch.randn(C) for k in ('img_channel_weights', 'pts_channel_weights')}
    swp = {k: torch.randn(Nq) for k in ('img_spatial_weights', 'pts_spatial_weights')}
    s = 2 if method == 'cat' else 1
    cot = torch.randn(Nq, B, C * s)

    def factors(P):
        """host-side composition exactly as UniBEVTransformer._channel_factors / _spatial_factors"""
        c, l = float(c_flag), float(l_flag)
        if method == 'avg':
            c, l = c / (c_flag + l_flag), l / (c_flag + l_flag)
        if norm == 'ChannelNormWeights':
            fw = torch.stack((P['img_channel_weights'], P['pts_channel_weights']), 0)
            if c_flag == 1 and l_flag == 1:
                n = fw.softmax(0)
                iw, pw = n[0], n[1]
            else:
                iw, pw = fw[0:1].softmax(0)[0], fw[1:2].softmax(0)[0]
            cw = (iw * c, pw * l)
        else:
            one = torch.ones(C, device=P['img_channel_weights'].device)
            cw = (one * c, one * l)
        sw = (None, None)
        if spatial:
            w = torch.stack((P['img_spatial_weights'], P['pts_spatial_weights']), 0)
            n = w.softmax(0) if (c_flag == 1 and l_flag == 1) else None
            sw = (n[0], n[1]) if n is not None else (w[:1].softmax(0)[0], w[1:].softmax(0)[0])
        return cw, sw

    Pg = {k: v.to(DEV).requires_grad_() for k, v in {**cwp, **swp}.items()}
    i1, p1 = img.clone().requires_grad_(), pts.clone().requires_grad_()
    cw, sw = factors(Pg)
    out = bev_fuse(i1, p1, cw[0], cw[1], sw[0], sw[1], cat=(method == 'cat'))
    (out * cot.to(DEV)).sum().backward()

    Pc = {k: v.clone().requires_grad_() for k, v in {**cwp, **swp}.items()}
    i2, p2 = img.cpu().clone().requires_grad_(), pts.cpu().clone().requires_grad_()
    a, b = R.channel_feature_norm(Pc, i2, p2, norm, c_flag, l_flag)
    a, b = R.spatial_feature_norm(Pc, a, b, 'SpatialNormWeights' if spatial else None, c_flag, l_flag)
    ref = R.multi_modal_fusion(a, b, method, c_flag, l_flag).permute(1, 0, 2)
    (ref * cot).sum().backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(i1.grad.cpu(), i2.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(p1.grad.cpu(), p2.grad, rtol=1e-5, atol=1e-6)
    if norm:
        for k in cwp:
            torch.testing.assert_close(Pg[k].grad.cpu(), Pc[k].grad, rtol=1e-3, atol=1e-3)
    if spatial:
        for k in swp:
            torch.testing.assert_close(Pg[k].grad.cpu(), Pc[k].grad, rtol=1e-3, atol=1e-3)


def test_bev_fuse_missing_modality_and_bad_shapes():
    from unibev_amd.functional import bev_fuse
    from unibev_amd._lib import UniBEVHipError
    x = torch.randn(1, 50, 64, device=DEV)
    one = torch.ones(64, device=DEV)
    out = bev_fuse(x, None, one, one, cat=True)
    assert out.shape == (50, 1, 128)
    torch.testing.assert_close(out[:, 0, :64], x[0])
    assert torch.all(out[:, 0, 64:] == 0)
    with pytest.raises(UniBEVHipError):
        bev_fuse(torch.randn(1, 5, 6, device=DEV), None, torch.ones(6, device=DEV),
                 torch.ones(6, device=DEV))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('rows,N,K', [(80000, 256, 256), (4099, 96, 256), (12345, 192, 64), (7, 512, 256)])
def test_linear_grad_reduce_vs_torch(dtype, rows, N, K):
    """Bias column sums and the split-K slice sum of a Linear's backward in one launch
    (ubv_linear_grad_reduce) against torch reductions in f64."""
    from unibev_amd.functional import linear_grad_reduce
    torch.manual_seed(rows)
    go = torch.randn(rows, N, device=DEV).to(dtype)
    part = torch.randn(5, N, K, device=DEV).to(dtype)
    gb, gw = linear_grad_reduce(go, part)
    assert gb.dtype == torch.float32 and gw.dtype == torch.float32 and gw.shape == (N, K)
    scale = max(1.0, rows ** 0.5)
    torch.testing.assert_close(gb.double(), go.double().sum(0), rtol=1e-5, atol=2e-5 * scale)
    torch.testing.assert_close(gw.double(), part.double().sum(0), rtol=1e-6, atol=1e-5)
    gb2, none = linear_grad_reduce(go, None)
    assert none is None
    torch.testing.assert_close(gb2, gb, rtol=1e-5, atol=1e-4 * scale)      # atomics: order varies
    none, gw2 = linear_grad_reduce(None, part)
    assert none is None and torch.equal(gw2, gw)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_function_gradients_vs_torch(dtype):
    """unibev_amd.linear.linear / linear_cat (library GEMMs + split-K weight gradient + fused
    reductions) against F.linear, forward and all gradients."""
    from unibev_amd.linear import linear, linear_cat
    torch.manual_seed(3)
    rows, K = 8192, 256
    x = torch.randn(2, rows // 2, K, device=DEV)
    l1, l2 = torch.nn.Linear(K, 128).to(DEV), torch.nn.Linear(K, 64).to(DEV)
    cot = torch.randn(2, rows // 2, 192, device=DEV)
    xa = x.clone().requires_grad_()
    with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
        y = linear_cat(xa, (l1.weight, l2.weight), (l1.bias, l2.bias))
    y.backward(cot.to(y.dtype))
    got = (y.detach().float(), xa.grad, l1.weight.grad, l2.weight.grad, l1.bias.grad, l2.bias.grad)
    for l in (l1, l2):
        l.weight.grad = l.bias.grad = None
    xr = x.clone().requires_grad_()
    yr = torch.cat((torch.nn.functional.linear(xr, l1.weight, l1.bias),
                    torch.nn.functional.linear(xr, l2.weight, l2.bias)), -1)
    yr.backward(cot)
    ref = (yr.detach(), xr.grad, l1.weight.grad, l2.weight.grad, l1.bias.grad, l2.bias.grad)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(got, ref):
        assert a.dtype == b.dtype
        torch.testing.assert_close(a, b, rtol=tol, atol=tol * float(b.abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_relu_dropout_fused(dtype):
    """dropout(relu(x)) in one pass: eval mode == relu forward and backward; train mode keeps
    ~(1-p) of the positive elements scaled by 1/(1-p), and the gradient follows the same mask
    (recovered from the output alone)."""
    from unibev_amd.functional import relu_dropout
    torch.manual_seed(0)
    x = torch.randn(4096, 512, device=DEV).to(dtype)
    cot = torch.randn_like(x)
    xa = x.clone().requires_grad_()
    y = relu_dropout(xa, 0.3, training=False)
    y.backward(cot)
    assert torch.equal(y.detach(), torch.relu(x))
    assert torch.equal(xa.grad, torch.where(x > 0, cot, torch.zeros_like(cot)))
    xb = x.clone().requires_grad_()
    p = 0.25
    y2 = relu_dropout(xb, p, training=True)
    y2.backward(cot)
    pos = x > 0
    kept = (y2 != 0) & pos
    frac = kept.float().sum() / pos.float().sum()
    assert abs(float(frac) - (1 - p)) < 0.01, float(frac)
    assert not bool((y2 != 0)[~pos].any())
    scale = 1.0 / (1.0 - p)
    torch.testing.assert_close(y2.detach()[kept].float(), (x[kept].float() * scale), rtol=1e-2, atol=1e-6)
    torch.testing.assert_close(xb.grad[kept].float(), cot[kept].float() * scale, rtol=1e-2, atol=1e-6)
    assert not bool(xb.grad[~kept].any())


def test_zero_arena_hands_out_disjoint_zeroed_slices():
    from unibev_amd import functional as UF
    UF.new_step()
    a = UF.zeros_f32(256, torch.device(DEV, 0))
    b = UF.zeros_f32(100, torch.device(DEV, 0))
    a += 1
    assert float(b.abs().sum()) == 0 and a.data_ptr() % 256 == 0 and b.data_ptr() % 256 == 0
    UF.new_step()
    c = UF.zeros_f32(256, torch.device(DEV, 0))
    assert float(c.abs().sum()) == 0 and float(a.sum()) == 256     # the old step's slices survive
    big = UF.zeros_f32(1 << 20, torch.device(DEV, 0))                # larger than the arena: regrows
    assert float(big.abs().sum()) == 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_pass_through_alias_gradients(dtype):
    """linear_pass(x) returns (y, x'): using x' for the residual must give the same values and the
    same gradients as using x itself (the two input gradients meet inside the input-gradient GEMM
    instead of a separate add)."""
    from unibev_amd.linear import linear, linear_pass
    torch.manual_seed(5)
    x = torch.randn(2, 4096, 256, device=DEV).to(dtype)
    lin = torch.nn.Linear(256, 256).to(DEV)
    c1, c2 = torch.randn_like(x), torch.randn_like(x)
    res = []
    for use_pass in (True, False):
        xa = x.clone().requires_grad_()
        lin.weight.grad = lin.bias.grad = None
        with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
            if use_pass:
                y, xr = linear_pass(xa, lin.weight, lin.bias)
            else:
                y, xr = linear(xa, lin.weight, lin.bias), xa
        assert torch.equal(xr.detach(), x)
        ((y.to(dtype) * c1).sum() + (xr * c2).sum()).backward()
        res.append((y.detach(), xa.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b in zip(*res):
        torch.testing.assert_close(a.float(), b.float(), rtol=tol, atol=tol * float(b.float().abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bias_gradient_rides_on_layernorm_backward(dtype):
    """Linear -> fused add+LayerNorm: the LayerNorm backward hands the column sums of grad_x to the
    Linear's backward as its bias gradient (no second pass over grad_x).  All gradients equal the
    torch composition."""
    from unibev_amd.functional import add_dropout_layernorm
    from unibev_amd.linear import linear
    torch.manual_seed(7)
    R, K, C = 6000, 128, 256
    x = torch.randn(R, K, device=DEV)
    idn = torch.randn(R, C, device=DEV)
    lin = torch.nn.Linear(K, C).to(DEV)
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.2, 0.2)
    cot = torch.randn(R, C, device=DEV)
    with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
        idt = idn.to(dtype) if dtype != torch.float32 else idn
        y = add_dropout_layernorm(linear(x, lin.weight, lin.bias), idt, ln.weight, ln.bias, 0.0, False)
    (y.float() * cot).sum().backward()
    got = [p.grad.clone() for p in (lin.weight, lin.bias, ln.weight, ln.bias)]
    for p in (lin.weight, lin.bias, ln.weight, ln.bias):
        p.grad = None
    yr = torch.nn.functional.layer_norm(torch.nn.functional.linear(x, lin.weight, lin.bias) + idn,
                                        (C,), ln.weight, ln.bias)
    (yr * cot).sum().backward()
    ref = [p.grad for p in (lin.weight, lin.bias, ln.weight, ln.bias)]
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=tol, atol=tol * float(b.abs().max()))
