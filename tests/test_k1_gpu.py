"""k1 (ms_deform_attn) on the GPU through the C ABI vs the oracle / reference-recorded vectors."""
import numpy as np
import pytest
import torch

from _util import golden, t
import make_golden as mg

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _levels(shapes, dev):
    ss = torch.as_tensor(np.asarray(shapes).reshape(-1, 2), dtype=torch.long, device=dev)
    ls = torch.cat((ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]))
    return ss, ls


@pytest.mark.parametrize('case', [c[0] for c in mg.MSDA_CASES])
def test_k1_fp32_matches_reference_vectors(case):
    """fp32 kernel vs the vectors recorded from the reference's CPU definition; tolerance 2e-5
    abs/rel on forward, 1e-4 on gradients (fp32 accumulation order differs)."""
    from unibev_amd.functional import ms_deform_attn
    g = golden('msda')
    v = t(g[f'{case}_value'], device=DEV).requires_grad_()
    l = t(g[f'{case}_loc'], device=DEV).requires_grad_()
    w = t(g[f'{case}_w'], device=DEV).requires_grad_()
    ss, ls = _levels(g[f'{case}_shapes'], DEV)
    out = ms_deform_attn(v, ss, ls, l, w, 64)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f'{case}_out64'], rtol=2e-5, atol=2e-5)
    out.backward(t(g[f'{case}_gout'], device=DEV))
    np.testing.assert_allclose(v.grad.cpu().numpy(), g[f'{case}_gvalue'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(l.grad.cpu().numpy(), g[f'{case}_gloc'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(w.grad.cpu().numpy(), g[f'{case}_gw'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize('case', ['c0', 'c1', 'c2'])
def test_k1_half_precision_within_tolerance(case, dtype, tol):
    """north_star bar: within 1e-3 rel (fp16) of the fp32 reference, measured as max |err| / max |ref|;
    bf16 (8 mantissa bits) is held to 8e-3."""
    from unibev_amd.functional import ms_deform_attn
    g = golden('msda')
    v = t(g[f'{case}_value'], device=DEV).to(dtype)
    ss, ls = _levels(g[f'{case}_shapes'], DEV)
    out = ms_deform_attn(v, ss, ls, t(g[f'{case}_loc'], device=DEV), t(g[f'{case}_w'], device=DEV))
    assert out.dtype == dtype
    ref = g[f'{case}_out64']
    err = np.abs(out.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < tol, err


@pytest.mark.parametrize('B,shapes,H,Dh,P,Nq', [
    (2, [(13, 17)], 8, 32, 4, 301),
    (1, [(20, 20)], 8, 16, 8, 400),
    (3, [(5, 7), (3, 4)], 8, 32, 4, 33),
    (1, [(6, 6)], 4, 32, 3, 10),        # H*LP = 32: two queries per wave, odd P
    (2, [(4, 5)], 3, 12, 2, 7),         # no lane tiling possible: fallback kernel
    (1, [(9, 9)], 8, 64, 4, 11),
])
def test_k1_random_shapes_vs_c_oracle(B, shapes, H, Dh, P, Nq):
    from oracle import c_ref
    from unibev_amd.functional import ms_deform_attn
    rs = np.random.RandomState(B * 1000 + Nq)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rs.standard_normal((B, S, H, Dh)).astype(np.float32)
    loc = (0.5 + 0.45 * rs.standard_normal((B, Nq, H, L, P, 2))).astype(np.float32)
    aw = rs.random_sample((B, Nq, H, L, P)).astype(np.float32)
    gout = rs.standard_normal((B, Nq, H * Dh)).astype(np.float32)
    ref = c_ref.msda_forward(value, shapes, loc, aw)
    gv, gl, gw = c_ref.msda_backward(value, shapes, loc, aw, gout)
    v = t(value, device=DEV).requires_grad_()
    l = t(loc, device=DEV).requires_grad_()
    w = t(aw, device=DEV).requires_grad_()
    ss, ls = _levels(shapes, DEV)
    out = ms_deform_attn(v, ss, ls, l, w)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    out.backward(t(gout, device=DEV))
    np.testing.assert_allclose(v.grad.cpu().numpy(), gv, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(l.grad.cpu().numpy(), gl, rtol=1e-4, atol=5e-4)
    np.testing.assert_allclose(w.grad.cpu().numpy(), gw, rtol=1e-4, atol=1e-4)


def test_k1_edge_locations():
    """Exactly-on-boundary, far outside, NaN and huge locations contribute what the definition
    says (zero outside; NaN locations are 'outside')."""
    from oracle import c_ref
    from unibev_amd.functional import ms_deform_attn
    shapes = [(4, 6)]
    value = np.arange(24 * 8 * 16, dtype=np.float32).reshape(1, 24, 8, 16) / 100.0
    pts = np.array([[0.0, 0.0], [1.0, 1.0], [-1.0 / 12, 0.5], [1.0 + 1.0 / 12, 0.5], [0.5, -0.126],
                    [5.0, 5.0], [-7.0, 0.3], [1e30, 0.5], [0.5 / 6, 0.5 / 4], [0.999999, 0.000001]],
                   np.float32)
    Nq = len(pts)
    loc = np.broadcast_to(pts[None, :, None, None, None, :], (1, Nq, 8, 1, 1, 2)).copy()
    aw = np.ones((1, Nq, 8, 1, 1), np.float32)
    ref = c_ref.msda_forward(value, shapes, loc, aw)
    ss, ls = _levels(shapes, DEV)
    out = ms_deform_attn(t(value, device=DEV), ss, ls, t(loc, device=DEV), t(aw, device=DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    loc[0, 0] = np.nan
    out = ms_deform_attn(t(value, device=DEV), ss, ls, t(loc, device=DEV), t(aw, device=DEV))
    assert torch.all(out[0, 0] == 0) and torch.isfinite(out).all()


def test_k1_empty_query_set_and_bad_args():
    from unibev_amd.functional import ms_deform_attn
    ss, ls = _levels([(4, 4)], DEV)
    v = torch.randn(1, 16, 8, 32, device=DEV)
    out = ms_deform_attn(v, ss, ls, torch.zeros(1, 0, 8, 1, 4, 2, device=DEV),
                         torch.zeros(1, 0, 8, 1, 4, device=DEV))
    assert out.shape == (1, 0, 256)
    with pytest.raises(ValueError):
        ms_deform_attn(v, ss, ls, torch.zeros(1, 3, 4, 1, 4, 2, device=DEV),
                       torch.zeros(1, 3, 8, 1, 4, device=DEV))
    with pytest.raises(RuntimeError):
        ms_deform_attn(v.cpu(), ss, ls, torch.zeros(1, 3, 8, 1, 4, 2), torch.zeros(1, 3, 8, 1, 4))


def test_k1_full_size_properties():
    """BASELINE sizes (self-attn instance: S = Nq = 40 000, H = 8, Dh = 32, P = 4): properties
    that hold at any size — linearity in value, weights summing to one reproduce a constant map,
    agreement with the C oracle on a strided sample of queries."""
    from oracle import c_ref
    from unibev_amd.functional import ms_deform_attn
    torch.manual_seed(0)
    B, Hq, Wq, H, Dh, P = 1, 200, 200, 8, 32, 4
    S = Nq = Hq * Wq
    ss, ls = _levels([(Hq, Wq)], DEV)
    ys, xs = torch.meshgrid(torch.arange(Hq, device=DEV), torch.arange(Wq, device=DEV), indexing='ij')
    ref = torch.stack(((xs + 0.5) / Wq, (ys + 0.5) / Hq), -1).view(1, Nq, 1, 1, 1, 2)
    loc = (ref + 0.02 * torch.randn(B, Nq, H, 1, P, 2, device=DEV)).contiguous()
    aw = torch.softmax(torch.randn(B, Nq, H, 1, P, device=DEV), -1)
    v1 = torch.randn(B, S, H, Dh, device=DEV)
    v2 = torch.randn(B, S, H, Dh, device=DEV)
    o1 = ms_deform_attn(v1, ss, ls, loc, aw)
    o2 = ms_deform_attn(v2, ss, ls, loc, aw)
    o12 = ms_deform_attn(2.0 * v1 - 3.0 * v2, ss, ls, loc, aw)
    torch.testing.assert_close(o12, 2.0 * o1 - 3.0 * o2, rtol=1e-4, atol=1e-4)
    inner = loc.clamp(0.01, 0.99)
    const = ms_deform_attn(torch.full_like(v1, 1.5), ss, ls, inner, aw)
    torch.testing.assert_close(const, torch.full_like(const, 1.5), rtol=1e-5, atol=1e-5)
    idx = torch.arange(0, Nq, 997, device=DEV)
    sub = c_ref.msda_forward(v1.cpu().numpy(), [(Hq, Wq)], loc[:, idx].cpu().numpy(),
                             aw[:, idx].cpu().numpy())
    np.testing.assert_allclose(o1[:, idx].cpu().numpy(), sub, rtol=2e-5, atol=2e-5)


def _hw_levels(shapes, dev):
    """spatial_shapes carrying its host copy, as the modules' ``shapes_tensor`` produces it: the backward may then
    plan owner tiles (``ubv_ms_deform_attn_backward_planned``)."""
    ss, ls = _levels(shapes, dev)
    ss._ubv_hw = [tuple(int(v) for v in s) for s in shapes]
    return ss, ls


@pytest.mark.parametrize('B,shape,H,Dh,P,Nq,spread', [
    (2, (13, 17), 8, 32, 4, 301, 0.45),
    (1, (20, 20), 8, 16, 8, 400, 0.45),
    (2, (9, 40), 8, 32, 8, 777, 0.3),
    (1, (16, 16), 8, 32, 4, 3000, 0.01),     # every point lands on a few pixels: bucket overflow path
])
def test_k1_planned_backward_vs_c_oracle_and_the_atomic_kernel(B, shape, H, Dh, P, Nq, spread):
    """Operator-level backward without grad_value atomics (owner tiles): same gradients as the C oracle within the
    atomic kernel's tolerances, and the two device paths agree."""
    from oracle import c_ref
    from unibev_amd import functional as UF
    rs = np.random.RandomState(Nq)
    S = shape[0] * shape[1]
    value = rs.standard_normal((B, S, H, Dh)).astype(np.float32)
    loc = (0.5 + spread * rs.standard_normal((B, Nq, H, 1, P, 2))).astype(np.float32)
    aw = rs.random_sample((B, Nq, H, 1, P)).astype(np.float32)
    gout = rs.standard_normal((B, Nq, H * Dh)).astype(np.float32)
    gv, gl, gw = c_ref.msda_backward(value, [shape], loc, aw, gout)
    got = {}
    for tag, levels in (('planned', _hw_levels), ('atomic', _levels)):
        v = t(value, device=DEV).requires_grad_()
        l = t(loc, device=DEV).requires_grad_()
        w = t(aw, device=DEV).requires_grad_()
        ss, ls = levels([shape], DEV)
        UF.enable_profile(True)
        UF.ms_deform_attn(v, ss, ls, l, w).backward(t(gout, device=DEV))
        prof = UF.profile_results()
        UF.enable_profile(False)
        assert ('k1_bwd_planned' in prof) == (tag == 'planned') and ('k1_bwd' in prof) == (tag == 'atomic'), list(prof)
        got[tag] = (v.grad, l.grad, w.grad)
        scale = np.abs(gv).max()
        np.testing.assert_allclose(v.grad.cpu().numpy(), gv, rtol=1e-4, atol=1e-4 * max(1.0, scale))
        np.testing.assert_allclose(l.grad.cpu().numpy(), gl, rtol=1e-4, atol=5e-4)
        np.testing.assert_allclose(w.grad.cpu().numpy(), gw, rtol=1e-4, atol=1e-4)
    # same query-side kernel in both (compiled with / without the atomic scatter: fused-multiply-add placement may differ)
    torch.testing.assert_close(got['planned'][1], got['atomic'][1], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got['planned'][2], got['atomic'][2], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-4), (torch.bfloat16, 1.5e-2)])
def test_k1_full_size_gradients_vs_c_oracle(dtype, tol):
    """BASELINE size (S = Nq = 40 000, H = 8, Dh = 32, P = 4): the planned backward's grad_value on the WHOLE map and
    grad_loc / grad_weight on a strided sample of queries against the f64 C oracle (normwise)."""
    from oracle import c_ref
    from unibev_amd.functional import ms_deform_attn
    rs = np.random.RandomState(7)
    B, Hq, Wq, H, Dh, P = 1, 200, 200, 8, 32, 4
    S = Nq = Hq * Wq
    ys, xs = np.meshgrid(np.arange(Hq), np.arange(Wq), indexing='ij')
    ref = np.stack(((xs + 0.5) / Wq, (ys + 0.5) / Hq), -1).reshape(1, Nq, 1, 1, 1, 2)
    loc = (ref + 0.015 * rs.standard_normal((B, Nq, H, 1, P, 2))).astype(np.float32)
    e = np.exp(rs.standard_normal((B, Nq, H, 1, P)))
    aw = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    value = rs.standard_normal((B, S, H, Dh)).astype(np.float32)
    gout = rs.standard_normal((B, Nq, H * Dh)).astype(np.float32)
    if dtype != torch.float32:                       # exactly representable operands
        value = t(value).to(dtype).float().numpy()
        gout = t(gout).to(dtype).float().numpy()
    gv, gl, gw = c_ref.msda_backward(value, [(Hq, Wq)], loc, aw, gout)
    v = t(value, device=DEV).to(dtype).requires_grad_()
    l = t(loc, device=DEV).requires_grad_()
    w = t(aw, device=DEV).requires_grad_()
    ss, ls = _hw_levels([(Hq, Wq)], DEV)
    ms_deform_attn(v, ss, ls, l, w).backward(t(gout, device=DEV).to(dtype))
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(v.grad.float().cpu().numpy(), gv) < tol
    idx = np.arange(0, Nq, 97)
    assert rel(l.grad.cpu().numpy()[:, idx], gl[:, idx]) < max(tol, 2e-4)
    assert rel(w.grad.cpu().numpy()[:, idx], gw[:, idx]) < max(tol, 2e-4)


@pytest.mark.parametrize('B,fh,fw,qh,qw,P,sig', [
    (2, 37, 41, 40, 45, 8, 0.04),      # partial edge tiles of the query grid and of the pixel grid
    (1, 50, 50, 50, 50, 4, 0.02),
    (2, 16, 16, 19, 13, 4, 0.6),       # locations far outside the map, boxes larger than the window
    (1, 180, 180, 200, 200, 8, 0.01),  # the SCA-pts instance's size
])
def test_k1_with_query_grid_hint_equals_the_plain_operator(B, fh, fw, qh, qw, P, sig):
    """``ubv_ms_deform_attn_forward_grid`` / ``_backward_grid`` (the operator on the TILE plan: the caller states that its
    queries are a qh x qw grid) against the plain operator on the same explicit locations / weights — forward to f32
    round-off, grad_value / grad_loc / grad_weight likewise (the owner tiles sum in another order than the atomics) —
    and against the fp64 oracle at the small sizes."""
    from unibev_amd import functional as UF
    from unibev_amd.modules.deform_attn import index_tensor, shapes_tensor
    from oracle import unibev_ref as R
    H, Dh = 8, 32
    Nq, S = qh * qw, fh * fw
    g = torch.Generator(device='cpu').manual_seed(B * 100 + P + qh)
    ys, xs = torch.meshgrid(torch.arange(qh), torch.arange(qw), indexing='ij')
    ref = torch.stack(((xs + 0.5) / qw, (ys + 0.5) / qh), -1).view(1, Nq, 1, 1, 1, 2)
    loc0 = ref + sig * torch.randn(B, Nq, H, 1, P, 2, generator=g)
    aw0 = torch.softmax(torch.randn(B, Nq, H, 1, P, generator=g), -1)
    v0 = torch.randn(B, S, H, Dh, generator=g)
    go = torch.randn(B, Nq, H * Dh, generator=g).to(DEV)
    ls = index_tensor([0], DEV)
    outs = {}
    for name, ss in (('plain', shapes_tensor([(fh, fw)], DEV)), ('grid', shapes_tensor([(fh, fw)], DEV, query_grid=(qh, qw)))):
        v, loc, aw = (x.clone().to(DEV).requires_grad_() for x in (v0, loc0, aw0))
        UF.kernel_profile(True)
        out = UF.ms_deform_attn(v, ss, ls, loc, aw)
        out.backward(go)
        torch.cuda.synchronize()
        kernels = set(UF.kernel_profile())
        UF.kernel_profile(False)
        assert any(k.startswith('k1_tile_') for k in kernels) == (name == 'grid'), kernels
        outs[name] = (out.detach(), v.grad, loc.grad, aw.grad)
    for a, b, what in zip(outs['grid'], outs['plain'], ('out', 'grad_value', 'grad_loc', 'grad_weight')):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) < 3e-5 * scale + 1e-6, (what, float((a - b).abs().max()), scale)
    if Nq <= 2500:
        v64, l64, w64 = (x.double().requires_grad_() for x in (v0, loc0, aw0))
        o64 = R.msda(v64, [(fh, fw)], l64, w64)
        o64.backward(go.cpu().double())
        for a, b, what in zip(outs['grid'], (o64.detach(), v64.grad, l64.grad, w64.grad), ('out', 'gv', 'gloc', 'gw')):
            a, b = a.cpu().double().numpy(), b.numpy()
            bad = np.abs(a - b) > 2e-4 * np.abs(b) + 2e-4 * max(float(np.abs(b).max()), 1.0)
            # d / d(location) jumps where a sample sits on a pixel boundary: f32 and f64 may floor() a location that
            # close to an integer to different sides — a handful of the 10^5 points, never the other tensors
            assert bad.mean() <= (2e-5 if what == 'gloc' else 0.0), (what, int(bad.sum()), bad.size)
