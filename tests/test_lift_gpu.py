"""Fused BEV lifting (ubv_bev_lift_*) vs the oracle's unfused composition (offsets -> locations,
softmax, k1, camera scatter-add, count divide), forward and backward."""
import numpy as np
import pytest
import torch

from _util import t
from oracle import unibev_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def oracle_lift(value, offlog, ref, vis0, count, Nc, fh, fw, H, P):
    """value (B*Nc, S, C), offlog (B, Nq, H*P*3), ref (Nc, B, Nq, Z, 2) — CPU fp64 composition
    following MSDeformableAttention3DImg.forward + SpatialCrossAttentionImg's accumulation."""
    BNc, S, C = value.shape
    B = BNc // Nc
    Nq = offlog.shape[1]
    Z = ref.shape[3]
    off = offlog[..., :H * P * 2].reshape(B, Nq, H, 1, P, 2)
    aw = offlog[..., H * P * 2:].reshape(B, Nq, H, P).softmax(-1).view(B, Nq, H, 1, P)
    norm = torch.tensor([fw, fh], dtype=value.dtype)
    off = (off / norm).view(B, Nq, H, 1, P // Z, Z, 2)
    out = torch.zeros(B, Nq, C, dtype=value.dtype)
    v = value.view(B, Nc, S, H, C // H)
    for cam in range(Nc):
        loc = (ref[cam][:, :, None, None, None, :, :] + off).view(B, Nq, H, 1, P, 2)
        o = R.msda(v[:, cam], [(fh, fw)], loc, aw)
        if vis0 is not None:
            o = o * vis0[cam].to(value.dtype)[None, :, None]
        out = out + o
    if count is not None:
        out = out / count[..., None]
    return out


CASES = [
    # B, Nc, fh, fw, H, Dh, qh, qw, P, Z
    (2, 1, 9, 11, 8, 32, 10, 12, 8, 4),      # SCA-pts shape class
    (1, 1, 10, 12, 8, 32, 10, 12, 4, 1),     # self-attn shape class
    (2, 3, 4, 6, 8, 32, 7, 9, 8, 4),         # SCA-img: 3 cameras, visibility + count
    (2, 2, 5, 5, 8, 16, 6, 5, 8, 4),         # Dh = 16 (cat-128 config)
    (1, 1, 16, 16, 8, 16, 16, 16, 4, 1),
    (1, 6, 8, 22, 8, 32, 20, 20, 8, 4),
]


def make_case(case, seed, with_vis):
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
    rs = np.random.RandomState(seed)
    Nq, C, S = qh * qw, H * Dh, fh * fw
    value = rs.standard_normal((B * Nc, S, C))
    offlog = np.concatenate([rs.standard_normal((B, Nq, H * P * 2)) * 1.5,
                             rs.standard_normal((B, Nq, H * P))], -1)
    ref = 0.5 + 0.3 * rs.standard_normal((Nc, B, Nq, Z, 2))
    vis0 = count = None
    if with_vis:
        vis_b = rs.random_sample((Nc, B, Nq)) < 0.4
        vis0 = vis_b[:, 0].astype(np.uint8)
        count = np.maximum(vis_b.sum(0), 1).astype(np.float32)
    gout = rs.standard_normal((B, Nq, C))
    return value, offlog, ref, vis0, count, gout


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('tiled', [True, False])
def test_lift_fp32_forward_backward(case, tiled):
    from unibev_amd.functional import bev_lift
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
    value, offlog, ref, vis0, count, gout = make_case(case, 7, Nc > 1)
    # oracle in float64
    v64 = t(value).requires_grad_()
    ol64 = t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref), None if vis0 is None else t(vis0),
                        None if count is None else t(count).double(), Nc, fh, fw, H, P)
    o_ref.backward(t(gout))
    v = t(value, torch.float32, DEV).requires_grad_()
    ol = t(offlog, torch.float32, DEV).requires_grad_()
    out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P,
                   vis0=None if vis0 is None else t(vis0, device=DEV),
                   count=None if count is None else t(count, device=DEV),
                   query_grid=(qh, qw) if tiled else None)
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.detach().numpy(), rtol=3e-5, atol=3e-5)
    out.backward(t(gout, torch.float32, DEV))
    np.testing.assert_allclose(v.grad.cpu().numpy(), v64.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ol.grad.cpu().numpy(), ol64.grad.numpy(), rtol=2e-4, atol=5e-4)


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_lift_half_precision(dtype, tol):
    from unibev_amd.functional import bev_lift
    case = CASES[2]
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
    value, offlog, ref, vis0, count, gout = make_case(case, 9, True)
    o_ref = oracle_lift(t(value), t(offlog), t(ref), t(vis0), t(count).double(), Nc, fh, fw, H, P)
    v = t(value, dtype, DEV).requires_grad_()
    ol = t(offlog, torch.float32, DEV).requires_grad_()
    out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P,
                   vis0=t(vis0, device=DEV), count=t(count, device=DEV), query_grid=(qh, qw))
    assert out.dtype == dtype
    err = (out.float().cpu() - o_ref.float()).abs().max() / o_ref.abs().max()
    assert err < tol, float(err)
    out.backward(t(gout, dtype, DEV))
    assert v.grad.dtype == dtype and torch.isfinite(ol.grad).all()


def test_lift_equals_k1_composition_at_full_size():
    """200x200 queries into a 180x180 map (SCA-pts instance), fp32: the fused kernel and the k1
    operator fed with torch-computed locations / softmax agree to fp32 round-off."""
    from unibev_amd.functional import bev_lift, ms_deform_attn
    torch.manual_seed(1)
    B, fh, fw, H, Dh, P, Z, qh, qw = 1, 180, 180, 8, 32, 8, 4, 200, 200
    Nq, C = qh * qw, H * Dh
    value = torch.randn(B, fh * fw, C, device=DEV)
    offlog = torch.randn(B, Nq, H * P * 3, device=DEV)
    offlog[..., :H * P * 2] *= 3.0
    ys, xs = torch.meshgrid(torch.arange(qh, device=DEV), torch.arange(qw, device=DEV), indexing='ij')
    ref = torch.stack(((xs + 0.5) / qw, (ys + 0.5) / qh), -1).view(1, 1, Nq, 1, 2).expand(1, B, Nq, Z, 2)
    fused = bev_lift(value, offlog, ref, 1, (fh, fw), H, P, query_grid=(qh, qw))
    off = offlog[..., :H * P * 2].view(B, Nq, H, 1, P, 2) / torch.tensor([fw, fh], device=DEV)
    loc = (ref[0][:, :, None, None, None, :, :] + off.view(B, Nq, H, 1, P // Z, Z, 2)).view(B, Nq, H, 1, P, 2)
    aw = offlog[..., H * P * 2:].view(B, Nq, H, P).softmax(-1).view(B, Nq, H, 1, P)
    ss = torch.tensor([[fh, fw]], device=DEV)
    ls = torch.zeros(1, dtype=torch.long, device=DEV)
    comp = ms_deform_attn(value.view(B, -1, H, Dh), ss, ls, loc.contiguous(), aw.contiguous())
    torch.testing.assert_close(fused, comp, rtol=2e-5, atol=2e-5)


def grid_ref(B, qh, qw, Z):
    ys, xs = np.meshgrid(np.arange(qh), np.arange(qw), indexing='ij')
    r = np.stack(((xs + 0.5) / qw, (ys + 0.5) / qh), -1).reshape(1, 1, qh * qw, 1, 2).astype(np.float32)
    return np.broadcast_to(r, (1, B, qh * qw, Z, 2)).copy()


@pytest.mark.parametrize('case', [
    # B, fh, fw, H, Dh, qh, qw, P, Z, offset sigma (pixels)
    (2, 37, 41, 8, 32, 40, 45, 8, 4, 3.0),     # SCA-pts class, partial edge tiles
    (1, 50, 50, 8, 32, 50, 50, 4, 1, 2.0),     # self-attn class
    (2, 33, 20, 8, 32, 36, 23, 8, 4, 9.0),     # many points beyond the near radius: atomic fallback
    (1, 48, 64, 8, 16, 52, 70, 4, 1, 4.0),     # Dh = 16
    (1, 16, 16, 8, 32, 16, 16, 8, 4, 30.0),    # offsets larger than the map
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_lift_backward_owner_tiles_grid_mode(case, dtype):
    """grad_value by LDS owner tiles (BEV-grid references) == the fp64 oracle, for offsets inside
    and outside the near radius, and == the all-atomics kernel."""
    from unibev_amd.functional import bev_lift
    B, fh, fw, H, Dh, qh, qw, P, Z, sig = case
    rs = np.random.RandomState(11)
    Nq, C, S = qh * qw, H * Dh, fh * fw
    value = rs.standard_normal((B, S, C))
    # offsets scatter (sigma `sig` pixels) around a per-slot centre, like a trained
    # sampling_offsets layer: bias + W q
    center = (rs.standard_normal(H * P * 2) * 4.0).astype(np.float32)
    offlog = np.concatenate([center[None, None] + rs.standard_normal((B, Nq, H * P * 2)) * sig,
                             rs.standard_normal((B, Nq, H * P))], -1)
    ref = grid_ref(B, qh, qw, Z)
    gout = rs.standard_normal((B, Nq, C))
    if dtype != torch.float32:          # make the 16-bit inputs exactly representable
        value = t(value).to(dtype).double().numpy()
        gout = t(gout).to(dtype).double().numpy()
    v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref).double(), None, None, 1, fh, fw, H, P)
    o_ref.backward(t(gout))
    grads = {}
    for grid in (True, 'centred', False):
        v = t(value, dtype, DEV).requires_grad_()
        ol = t(offlog, torch.float32, DEV).requires_grad_()
        out = bev_lift(v, ol, t(ref, torch.float32, DEV), 1, (fh, fw), H, P, query_grid=(qh, qw),
                       ref_is_grid=bool(grid),
                       slot_center=t(center, device=DEV) if grid == 'centred' else None)
        out.backward(t(gout, dtype, DEV))
        grads[grid] = (v.grad.float().cpu().numpy(), ol.grad.cpu().numpy())
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    scale = np.abs(v64.grad.numpy()).max()
    for grid in (True, 'centred', False):
        np.testing.assert_allclose(grads[grid][0], v64.grad.numpy(), rtol=tol, atol=tol * scale)
        # d/d(offset) is discontinuous where a sampling point sits exactly on a pixel boundary
        # (grid-centred references put points within f32 round-off of one now and then): allow a
        # handful of such elements, everything else must agree
        bad = ~np.isclose(grads[grid][1], ol64.grad.numpy(), rtol=2e-4, atol=1e-3)
        assert bad.sum() <= 1e-5 * bad.size + 1, int(bad.sum())


def test_lift_backward_camera_mode_bands_and_chunks():
    """SCA-img maps of the three BASELINE sizes: 8x22 (one band), 25x45 and 29x50 (row bands),
    6 cameras, visibility + count; grad_value vs the fp64 oracle."""
    from unibev_amd.functional import bev_lift
    for (fh, fw, Dh) in [(8, 22, 32), (25, 45, 16), (29, 50, 32)]:
        case = (2, 6, fh, fw, 8, Dh, 30, 33, 8, 4)
        B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
        value, offlog, ref, vis0, count, gout = make_case(case, 13, True)
        v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
        o_ref = oracle_lift(v64, ol64, t(ref), t(vis0), t(count).double(), Nc, fh, fw, H, P)
        o_ref.backward(t(gout))
        v = t(value, torch.float32, DEV).requires_grad_()
        ol = t(offlog, torch.float32, DEV).requires_grad_()
        out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P,
                       vis0=t(vis0, device=DEV), count=t(count, device=DEV), query_grid=(qh, qw))
        out.backward(t(gout, torch.float32, DEV))
        scale = np.abs(v64.grad.numpy()).max()
        np.testing.assert_allclose(v.grad.cpu().numpy(), v64.grad.numpy(), rtol=2e-4, atol=2e-4 * scale)
        bad = ~np.isclose(ol.grad.cpu().numpy(), ol64.grad.numpy(), rtol=2e-4, atol=1e-3)
        assert bad.sum() <= 1e-5 * bad.size + 1, int(bad.sum())       # pixel-boundary discontinuities


@pytest.mark.parametrize('grid', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_lift_backward_maps_plan_heavy_buckets(dtype, grid):
    """Large per-camera maps (25x45: the MAPS plan, exact CSR buckets + work items).  All reference
    points sit in a 3-pixel patch, so a few buckets hold tens of thousands of records and are cut into
    many work items whose partial tiles are summed in order: grad_value vs the fp64 oracle, and two
    launches against each other."""
    from unibev_amd.functional import bev_lift
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = 1, 2, 25, 45, 8, 16, 48, 48, 8, 4
    rs = np.random.RandomState(3)
    Nq, C, S = qh * qw, H * Dh, fh * fw
    value = rs.standard_normal((B * Nc, S, C))
    offlog = np.concatenate([rs.standard_normal((B, Nq, H * P * 2)) * 1.5,
                             rs.standard_normal((B, Nq, H * P))], -1)
    ref = 0.45 + 0.06 * rs.random_sample((Nc, B, Nq, Z, 2))
    ref[1] += 0.3                                           # camera 1: a second patch
    vis0 = (rs.random_sample((Nc, Nq)) < 0.8)
    vis0[0, :64] = False                                     # a whole 8x8 query tile invisible in camera 0
    count = np.maximum(vis0.sum(0), 1).astype(np.float32)[None].repeat(B, 0)
    gout = rs.standard_normal((B, Nq, C))
    if dtype != torch.float32:
        value = t(value).to(dtype).double().numpy()
        gout = t(gout).to(dtype).double().numpy()
    v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref), t(vis0.astype(np.uint8)), t(count).double(), Nc, fh, fw, H, P)
    o_ref.backward(t(gout))
    got = []
    for _ in range(2):
        v = t(value, dtype, DEV).requires_grad_()
        ol = t(offlog, torch.float32, DEV).requires_grad_()
        out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P,
                       vis0=t(vis0.astype(np.uint8), device=DEV), count=t(count, device=DEV),
                       query_grid=(qh, qw) if grid else None)      # None: queries in list order (64 per wave)
        out.backward(t(gout, dtype, DEV))
        got.append(v.grad.clone())
    # records enter a bucket in arrival order (as on the GRID plan): launches agree to the order of the f32 sums
    torch.testing.assert_close(got[0].float(), got[1].float(), rtol=2.0 ** -7 if dtype != torch.float32 else 1e-4, atol=1e-3)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    scale = np.abs(v64.grad.numpy()).max()
    np.testing.assert_allclose(got[0].float().cpu().numpy(), v64.grad.numpy(), rtol=tol, atol=tol * scale)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('mode', ['grid', 'centred', 'atomics', 'camera'])
def test_lift_16bit_offsets_logits_read_directly(mode, dtype):
    """Under autocast the sampling_offsets / attention_weights Linear emits 16-bit rows; the
    kernels read them (and write their gradients) in that type.  Must be BIT-identical to
    up-casting the rows to f32 first and rounding the f32 gradients afterwards."""
    from unibev_amd.functional import bev_lift
    if mode == 'camera':
        case = (2, 6, 8, 22, 8, 32, 30, 33, 8, 4)
        B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
        value, offlog, ref, vis0, count, gout = make_case(case, 21, True)
        kw = dict(vis0=t(vis0, device=DEV), count=t(count, device=DEV), query_grid=(qh, qw))
    else:
        B, Nc, fh, fw, H, Dh, qh, qw, P, Z = 2, 1, 37, 41, 8, 32, 40, 45, 8, 4
        rs = np.random.RandomState(5)
        Nq = qh * qw
        value = rs.standard_normal((B, fh * fw, H * Dh))
        center = (rs.standard_normal(H * P * 2) * 4.0).astype(np.float32)
        offlog = np.concatenate([center[None, None] + rs.standard_normal((B, Nq, H * P * 2)) * 4.0,
                                 rs.standard_normal((B, Nq, H * P))], -1)
        ref = grid_ref(B, qh, qw, Z)
        gout = rs.standard_normal((B, Nq, H * Dh))
        kw = dict(query_grid=(qh, qw), ref_is_grid=mode != 'atomics',
                  slot_center=t(center, device=DEV) if mode == 'centred' else None)
    res = []
    for lowp in (True, False):
        v = t(value, dtype, DEV).requires_grad_()
        ol16 = t(offlog, dtype, DEV)
        ol = (ol16 if lowp else ol16.float()).requires_grad_()
        out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P, **kw)
        out.backward(t(gout, dtype, DEV))
        assert ol.grad.dtype == ol.dtype
        res.append((out.detach(), v.grad, ol.grad.to(dtype)))
    names = ('out', 'grad_value', 'grad_offlog')
    for name, a, b in zip(names, *res):
        if name == 'grad_value' and mode != 'camera':
            # far corners / the all-atomics plan add f32 atomically: the order varies run to run
            torch.testing.assert_close(a.float(), b.float(), rtol=2e-2, atol=2e-2)
        elif name == 'grad_offlog':
            # the two instantiations (16-bit / f32 rows) are compiled separately:
            # a differently contracted multiply-add may move a gradient by one f32 ulp, which now
            # and then crosses a 16-bit rounding boundary
            diff = (a != b)
            assert diff.float().mean().item() < 1e-4, int(diff.sum())
            torch.testing.assert_close(a.float(), b.float(), rtol=2.0 ** -7, atol=1e-6)
        else:
            assert torch.equal(a, b), name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_lift_backward_bins_overflow_is_exact(dtype):
    """GRID plan with every sampling point of every query aimed at the same few pixels: the owner
    tiles' buckets overflow by two orders of magnitude and the overflow list carries almost all of
    grad_value.  Still equal to the fp64 oracle."""
    from unibev_amd.functional import bev_lift
    B, fh, fw, H, Dh, qh, qw, P, Z = 2, 40, 36, 8, 32, 48, 40, 8, 4
    rs = np.random.RandomState(3)
    Nq, C, S = qh * qw, H * Dh, fh * fw
    value = rs.standard_normal((B, S, C))
    ref = grid_ref(B, qh, qw, Z)                                    # (1, B, Nq, Z, 2)
    target = np.array([0.37, 0.58])                                 # normalised map location
    # offsets (pixels) that move anchor p % Z of every query onto the target, +- 1.5 pixels
    delta = (target[None, None, None, :] - ref[0][:, :, [p % Z for p in range(P)], :]) * np.array([fw, fh])
    off = delta[:, :, None, :, :] + rs.uniform(-1.5, 1.5, (B, Nq, H, P, 2))
    offlog = np.concatenate([off.reshape(B, Nq, H * P * 2), rs.standard_normal((B, Nq, H * P))], -1)
    gout = rs.standard_normal((B, Nq, C))
    if dtype != torch.float32:
        value = t(value).to(dtype).double().numpy()
        gout = t(gout).to(dtype).double().numpy()
    v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref).double(), None, None, 1, fh, fw, H, P)
    o_ref.backward(t(gout))
    v = t(value, dtype, DEV).requires_grad_()
    ol = t(offlog, torch.float32, DEV).requires_grad_()
    out = bev_lift(v, ol, t(ref, torch.float32, DEV), 1, (fh, fw), H, P, query_grid=(qh, qw),
                   ref_is_grid=True)
    out.backward(t(gout, dtype, DEV))
    tol = 3e-4 if dtype == torch.float32 else 2e-2
    scale = np.abs(v64.grad.numpy()).max()
    np.testing.assert_allclose(v.grad.float().cpu().numpy(), v64.grad.numpy(), rtol=tol, atol=tol * scale)
    bad = ~np.isclose(ol.grad.cpu().numpy(), ol64.grad.numpy(), rtol=2e-4, atol=2e-3)
    assert bad.sum() <= 1e-5 * bad.size + 1, int(bad.sum())


@pytest.mark.parametrize('dtype,ftol', [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize('case', [
    # B, Nc, fh, fw, H, Dh, qh, qw, P, Z
    (2, 6, 8, 22, 8, 32, 30, 33, 8, 4),      # the BASELINE camera maps (176 pixels, 11 K-blocks)
    (1, 3, 4, 6, 8, 32, 7, 9, 8, 4),         # 24 pixels: 2 K-blocks, one partial
    (2, 2, 12, 16, 8, 32, 17, 20, 8, 2),     # 192 pixels: every K-block, both backward passes
    (1, 1, 6, 9, 4, 32, 11, 13, 8, 8),       # one camera, 4 heads, Z = 8
])
@pytest.mark.parametrize('tiled', [True, False])
def test_lift_camera_matrix_core_plan(case, dtype, ftol, tiled):
    """Small per-camera maps with 16-bit data run the gather as an MFMA product against a
    coefficient matrix built in LDS (bev_lift_cam.inl).  Inputs are made exactly representable, so
    the forward differs from the fp64 oracle only by the rounding of the coefficients / the
    output, and d(offsets), d(logits) — computed from f32 dot products — agree to f32 accuracy."""
    from unibev_amd.functional import bev_lift
    from unibev_amd._lib import lib, UBV_F16, UBV_BF16
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
    code = UBV_F16 if dtype == torch.float16 else UBV_BF16
    assert lib().ubv_bev_lift_forward_workspace(B, Nc, fh, fw, H, Dh, P, code) > 0
    value, offlog, ref, vis0, count, gout = make_case(case, 17, Nc > 1)
    value = t(value).to(dtype).double().numpy()
    gout = t(gout).to(dtype).double().numpy()
    v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref), None if vis0 is None else t(vis0),
                        None if count is None else t(count).double(), Nc, fh, fw, H, P)
    o_ref.backward(t(gout))
    v = t(value, dtype, DEV).requires_grad_()
    ol = t(offlog, torch.float32, DEV).requires_grad_()
    out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P,
                   vis0=None if vis0 is None else t(vis0, device=DEV),
                   count=None if count is None else t(count, device=DEV),
                   query_grid=(qh, qw) if tiled else None)
    err = (out.float().cpu() - o_ref.detach().float()).abs().max() / o_ref.abs().max()
    assert err < ftol, float(err)
    out.backward(t(gout, dtype, DEV))
    scale = np.abs(v64.grad.numpy()).max()
    np.testing.assert_allclose(v.grad.float().cpu().numpy(), v64.grad.numpy(), rtol=2e-2, atol=2e-2 * scale)
    bad = ~np.isclose(ol.grad.cpu().numpy(), ol64.grad.numpy(), rtol=2e-4, atol=1e-3)
    assert bad.sum() <= 1e-5 * bad.size + 1, int(bad.sum())       # pixel-boundary discontinuities


@pytest.mark.parametrize('case', [
    # B, Nc, fh, fw, H, Dh, qh, qw, P, Z
    (2, 6, 8, 22, 8, 32, 30, 33, 8, 4),      # the BASELINE camera maps (176 pixels, 14 K-blocks)
    (1, 3, 4, 6, 8, 32, 7, 9, 8, 4),         # 24 pixels
    (2, 2, 12, 16, 8, 32, 17, 20, 8, 2),     # 192 pixels: 15 K-blocks, 8 row blocks, every backward pass
    (1, 1, 6, 9, 4, 32, 11, 13, 8, 8),       # one camera, 4 heads, Z = 8
])
@pytest.mark.parametrize('tiled', [True, False])
def test_lift_camera_matrix_core_plan_fp32(case, tiled):
    """f32 data on the matrix-core CAMERA plan (bev_lift_cam32.inl): both MFMA operands split into bf16 hi + lo
    halves, three products per K-block, forward and both backward kernels.  Held to the f32 bars of test_lift_fp32_forward_backward against the fp64
    oracle — the split must not be visible."""
    from unibev_amd.functional import bev_lift
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
    value, offlog, ref, vis0, count, gout = make_case(case, 23, Nc > 1)
    v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref), None if vis0 is None else t(vis0),
                        None if count is None else t(count).double(), Nc, fh, fw, H, P)
    o_ref.backward(t(gout))
    v = t(value, torch.float32, DEV).requires_grad_()
    ol = t(offlog, torch.float32, DEV).requires_grad_()
    out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P,
                   vis0=None if vis0 is None else t(vis0, device=DEV),
                   count=None if count is None else t(count, device=DEV),
                   query_grid=(qh, qw) if tiled else None)
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.detach().numpy(), rtol=3e-5, atol=3e-5)
    out.backward(t(gout, torch.float32, DEV))
    np.testing.assert_allclose(v.grad.cpu().numpy(), v64.grad.numpy(), rtol=2e-4, atol=2e-4)
    bad = ~np.isclose(ol.grad.cpu().numpy(), ol64.grad.numpy(), rtol=2e-4, atol=5e-4)
    assert bad.sum() <= 1e-5 * bad.size + 1, int(bad.sum())       # pixel-boundary discontinuities


def test_lift_camera_matrix_core_plan_fp32_backward_only_mode():
    """UBV_CAM_MFMA32=2 keeps the f32 forward on the gather kernel and only the two backward kernels on the matrix
    cores (an A/B knob, read once per process) — the parity cases above are re-run that way in a child process."""
    import os
    import subprocess
    import sys
    if os.environ.get('UBV_CAM_MFMA32') == '2':
        pytest.skip('already the child run')
    env = dict(os.environ, UBV_CAM_MFMA32='2')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_lift_gpu.py'), '-q', '-x',
                        '-k', 'test_lift_camera_matrix_core_plan_fp32 and not backward_only'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '8 passed' in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize('plan', ['grid', 'camera'])
def test_lift_backward_repeatable_and_independent_of_kernel_concurrency(plan):
    """Round 1 saw lift_bwd_query_kernel return a handful of different values in 1-2 % of launches when
    the grad_value chain ran on a second stream beside it, and parked the two-stream variant
    unexplained.  Nothing in the kernels explains it (the query kernel reads only inputs and writes
    only its own rows; the grad_value chain shares no buffer with it); the removed host code did not
    order the side stream's use of the cached workspace / the allocator's reuse of freed gradients
    against the next call.  The variant rebuilt here forks and joins with events INSIDE the C call
    (UBV_LIFT_TWO_STREAM=1): over 250 launches each way, d(offsets) / d(logits) are bit-identical
    between launches and between the one- and two-stream schedules, on both owner-tile plans;
    grad_value is bit-identical on the CAMERA plan and agrees to the order of its f32 bucket sums on
    the GRID plan (records enter a bucket in arrival order)."""
    import os
    from unibev_amd.functional import bev_lift, compact_visible
    if plan == 'grid':
        B, Nc, fh, fw, H, Dh, qh, qw, P, Z = 2, 1, 90, 90, 8, 32, 100, 100, 8, 4
        rs = np.random.RandomState(2)
        Nq = qh * qw
        value = rs.standard_normal((B, fh * fw, H * Dh))
        offlog = np.concatenate([rs.standard_normal((B, Nq, H * P * 2)) * 3.0, rs.standard_normal((B, Nq, H * P))], -1)
        ref, gout = grid_ref(B, qh, qw, Z), rs.standard_normal((B, Nq, H * Dh))
        kw = dict(query_grid=(qh, qw), ref_is_grid=True)
    else:
        case = (2, 6, 8, 22, 8, 32, 60, 60, 8, 4)
        B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
        value, offlog, ref, vis0, count, gout = make_case(case, 23, True)
        v0 = t(vis0, device=DEV)
        kw = dict(vis0=v0, count=t(count, device=DEV), query_grid=(qh, qw), visible_lists=compact_visible(v0))
    dt = torch.bfloat16
    v = t(value, dt, DEV).requires_grad_()
    ol = t(offlog, dt, DEV).requires_grad_()
    r = t(ref, torch.float32, DEV)
    go = t(gout, dt, DEV)
    base = first = None
    try:
        for two in ('0', '1'):
            os.environ['UBV_LIFT_TWO_STREAM'] = two
            for it in range(250):
                v.grad = ol.grad = None
                bev_lift(v, ol, r, Nc, (fh, fw), H, P, **kw).backward(go)
                if base is None:
                    base = first = (ol.grad.clone(), v.grad.clone())
                    continue
                if plan == 'grid' and two == '1' and it == 0:
                    # Since round 5 the one-stream GRID backward of 16-bit maps is the TILE query kernel (it also bins, so
                    # there is no separate chain to put on a side stream); the two-stream diagnostic keeps the shared-footprint
                    # kernel + lift_bin_kernel.  Two kernels, two orders of the same f32 sums: equal to the rounding of the
                    # 16-bit result, each bit-identical to itself from then on.
                    torch.testing.assert_close(ol.grad.float(), first[0].float(), rtol=2.0 ** -6, atol=2e-3)
                    base = (ol.grad.clone(), v.grad.clone())
                    continue
                assert torch.equal(ol.grad, base[0]), two
                if plan == 'camera':
                    assert torch.equal(v.grad, base[1]), two
                else:
                    torch.testing.assert_close(v.grad.float(), base[1].float(), rtol=2.0 ** -7, atol=1e-3)
    finally:
        os.environ.pop('UBV_LIFT_TWO_STREAM', None)


def test_visible_lists_tile_by_tile_hold_the_same_queries_and_gradients():
    """ubv_compact_visible_grid: with the width of a BEV query grid of whole 8x8 tiles the per-camera lists come tile
    by tile (row-major inside a tile) — the same queries as the ascending lists, in the order of the tile-major walk —
    and the CAMERA backward that walks them gives the same gradients (a different order of its f32 partial sums);
    a grid that is not whole tiles keeps the ascending order."""
    from unibev_amd.functional import bev_lift, compact_visible
    case = (2, 6, 8, 22, 8, 32, 64, 64, 8, 4)
    B, Nc, fh, fw, H, Dh, qh, qw, P, Z = case
    value, offlog, ref, vis0, count, gout = make_case(case, 31, True)
    v0 = t(vis0, device=DEV)
    Nq = qh * qw
    asc, tiled = compact_visible(v0).cpu().numpy(), compact_visible(v0, qw).cpu().numpy()
    assert np.array_equal(asc[Nc * Nq:], tiled[Nc * Nq:])                       # the counts
    q = np.arange(Nq)
    rank = ((q // qw // 8) * (qw // 8) + (q % qw) // 8) * 64 + (q // qw % 8) * 8 + q % 8     # position in the walk
    for c in range(Nc):
        n = int(asc[Nc * Nq + c])
        a, b = asc[c * Nq:c * Nq + n], tiled[c * Nq:c * Nq + n]
        assert np.array_equal(a, np.flatnonzero(vis0[c]))
        assert np.array_equal(np.sort(b), a)
        assert np.all(np.diff(rank[b]) > 0)
    odd = compact_visible(v0[:, :60 * 60].contiguous(), 60).cpu().numpy()       # 60 is not a multiple of 8: ascending
    for c in range(Nc):
        n = int(odd[Nc * 3600 + c])
        assert np.all(np.diff(odd[c * 3600:c * 3600 + n]) > 0)
    grads = []
    for lists in (compact_visible(v0), compact_visible(v0, qw)):
        v = t(value, torch.float32, DEV).requires_grad_()
        ol = t(offlog, torch.float32, DEV).requires_grad_()
        out = bev_lift(v, ol, t(ref, torch.float32, DEV), Nc, (fh, fw), H, P, vis0=v0, count=t(count, device=DEV),
                       query_grid=(qh, qw), visible_lists=lists)
        out.backward(t(gout, torch.float32, DEV))
        grads.append((v.grad.clone(), ol.grad.clone()))
    assert torch.equal(grads[0][1], grads[1][1])                                # the query side does not walk the lists
    torch.testing.assert_close(grads[0][0], grads[1][0], rtol=1e-5, atol=1e-5 * float(grads[0][0].abs().max()))


@pytest.mark.parametrize('P,Z', [(4, 1), (8, 4)])
def test_tile_plan_with_every_point_off_the_map_over_poisoned_lds(P, Z):
    """a (tile, head) whose sampling points ALL fall outside the map has an empty pixel box, loads no window,
    and its weightless corners are read from window pixel (0, 0) and multiplied by 0 — NaN if the LDS left behind by an
    earlier kernel holds NaN bits.  Half of the tiles get offsets far off the map, the LDS of every CU is filled with
    NaNs first: outputs and gradients stay finite and equal the fp64 oracle (exactly 0 for the off-map queries)."""
    from unibev_amd import functional as UF
    from unibev_amd._lib import lib, check
    B, fh, fw, H, Dh, qh, qw = 1, 24, 24, 8, 32, 32, 32
    rs = np.random.RandomState(5)
    Nq, C = qh * qw, H * Dh
    value = rs.standard_normal((B, fh * fw, C))
    offs = rs.standard_normal((B, qh, qw, H * P * 2)) * 1.5
    offs[:, :, :16] = 1000.0 + 50.0 * rs.standard_normal((B, qh, 16, H * P * 2))        # left half: far off the map
    offlog = np.concatenate([offs.reshape(B, Nq, -1), rs.standard_normal((B, Nq, H * P))], -1)
    ref = grid_ref(B, qh, qw, Z)
    gout = rs.standard_normal((B, Nq, C))
    v64, ol64 = t(value).requires_grad_(), t(offlog).requires_grad_()
    o_ref = oracle_lift(v64, ol64, t(ref).double(), None, None, 1, fh, fw, H, P)
    o_ref.backward(t(gout))
    v = t(value, torch.float32, DEV).requires_grad_()
    ol = t(offlog, torch.float32, DEV).requires_grad_()
    nan_bits = 0x7fc00000
    for _ in range(2):
        check(lib().ubv_debug_fill_lds(nan_bits, UF._stream()), 'debug_fill_lds')
    out = UF.bev_lift(v, ol, t(ref, torch.float32, DEV), 1, (fh, fw), H, P, query_grid=(qh, qw))
    check(lib().ubv_debug_fill_lds(nan_bits, UF._stream()), 'debug_fill_lds')
    out.backward(t(gout, torch.float32, DEV))
    assert torch.isfinite(out).all() and torch.isfinite(ol.grad).all() and torch.isfinite(v.grad).all()
    off_map = out.detach().view(B, qh, qw, C)[:, :, :16]
    assert float(off_map.abs().max()) == 0.0
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.detach().numpy(), rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(v.grad.cpu().numpy(), v64.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ol.grad.cpu().numpy(), ol64.grad.numpy(), rtol=2e-4, atol=5e-4)


def test_experimental_tile_kernels_stay_parity_green():
    """The two round-6 kernels that measured slower and are OFF by default — query records for the TILE backward
    (UBV_LIFT_QREC=1, profiles/r06_qrec_experiment.txt) and the persistent pipelined TILE forward (UBV_TILE_PIPE=4, both
    forms; profiles/r06_tile_pipe_experiment.txt) — stay correct: the f32 forward / backward parity cases, the owner-tile
    cases (offsets up to 30 px on small maps: overflow and far-tile paths), the exact-overflow case and the poisoned-LDS
    case of this file, re-run in a process with the knobs on (they are read once per process)."""
    import os
    import subprocess
    import sys
    sel = 'fp32_forward_backward or owner_tiles_grid_mode or bins_overflow_is_exact or poisoned_lds or k1_composition'
    for extra in ({'UBV_LIFT_QREC': '1', 'UBV_TILE_PIPE': '4'}, {'UBV_TILE_PIPE': '3', 'UBV_TILE_PIPE_LIGHT': '0'}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-k', sel,
                            '-p', 'no:cacheprovider'], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (extra, r.stdout[-2000:])
        assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-500:]
