"""Modulated deformable convolution (``ubv_dcn_im2col`` / ``ubv_dcn_col2im`` + the MFMA GEMMs) and the DCN ResNet
stages on the GPU against the oracle (oracle/dcn_ref.py: mmcv's published kernel restated, unpinned — mmcv is not in
the reference tree).  Tolerances: f32 runs its GEMMs as split-bf16 products (2^-16 relative) and sums d(input) with
f32 atomics -> 2e-4 normwise; bf16 / fp16 data -> 2e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _case(seed, N, C, H, W, Cout, stride, pad, dil, dg, scale=1.5):
    g = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = torch.randn(N, C, H, W, generator=g)
    off = scale * torch.randn(N, dg * 18, Ho, Wo, generator=g)
    off[:, :, 0, 0] = 40.0                                   # far outside the map: the <= -1 / >= size rule
    off[:, 0, -1, -1] = -1.0 - 1.0                           # exactly on an integer position outside
    mask = torch.rand(N, dg * 9, Ho, Wo, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Cout, generator=g)
    return x, off, mask, w, b


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
@pytest.mark.parametrize('geom', [(2, 32, 11, 13, 64, 1, 1, 1, 1), (1, 64, 16, 12, 32, 2, 1, 1, 2),
                                  (2, 32, 9, 9, 32, 1, 2, 2, 4), (1, 256, 16, 44, 256, 1, 1, 1, 1)])
def test_function_forward_and_gradients_vs_oracle(geom, dtype, tol):
    from oracle import dcn_ref as R
    import unibev_amd.functional as UF
    N, C, H, W, Cout, stride, pad, dil, dg = geom
    x, off, mask, w, b = _case(3, *geom)
    ref_in = [t.clone().requires_grad_(True) for t in (x, off, mask, w, b)]
    # the oracle sees the operands as the device does (rounded to the data type)
    rx, ro, rm = (t.detach().to(dtype).float().requires_grad_(True) for t in (x, off, mask))
    rw = ref_in[3].detach().to(dtype).float().requires_grad_(True) if dtype != torch.float32 else ref_in[3]
    want = R.modulated_deform_conv2d(rx, ro, rm, rw, ref_in[4], stride, pad, dil, 1, dg)
    gy = torch.randn(want.shape, generator=torch.Generator().manual_seed(9))
    want.backward(gy)
    dx, do, dm = (t.to(DEV).to(dtype).requires_grad_(True) for t in (x, off, mask))
    dw, db = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    got = UF.modulated_deform_conv2d(dx, do, dm, dw, db, stride, pad, dil, 1, dg)
    assert got.shape == want.shape and got.dtype == dtype
    assert _rel(got, want) < tol
    got.backward(gy.to(DEV).to(dtype))
    assert _rel(dx.grad, rx.grad) < tol and _rel(dm.grad, rm.grad) < tol and _rel(do.grad, ro.grad) < 2 * tol
    assert _rel(dw.grad, rw.grad) < tol and _rel(db.grad, ref_in[4].grad) < tol


def test_zero_offsets_unit_mask_is_the_library_convolution_and_bad_arguments_fail():
    import unibev_amd.functional as UF
    from unibev_amd._lib import UniBEVHipError
    x = torch.randn(2, 32, 10, 14, device=DEV)
    w = torch.randn(64, 32, 3, 3, device=DEV) / 17
    off = torch.zeros(2, 18, 10, 14, device=DEV)
    m = torch.ones(2, 9, 10, 14, device=DEV)
    got = UF.modulated_deform_conv2d(x, off, m, w, None, 1, 1, 1, 1, 1)
    assert _rel(got, torch.nn.functional.conv2d(x, w, None, 1, 1)) < 1e-4
    with pytest.raises(UniBEVHipError):
        UF.modulated_deform_conv2d(x[:, :30], off, m, w[:, :30], None, 1, 1, 1, 1, 4)       # 30 / 4 channels per group
    with pytest.raises(NotImplementedError):
        UF.modulated_deform_conv2d(x, off, m, w, None, 1, 1, 1, 2, 1)
    with pytest.raises(RuntimeError):
        UF.modulated_deform_conv2d(x.cpu(), off.cpu(), m.cpu(), w.cpu(), None, 1, 1, 1, 1, 1)  # no CPU fallback


def test_pack_and_dcn_resnet_stage_train_on_the_device():
    from oracle import dcn_ref as R
    from unibev_amd.modules import FPN, ModulatedDeformConv2dPack, ResNet, extract_img_feat
    torch.manual_seed(0)
    pack = ModulatedDeformConv2dPack(32, 32, 3, stride=1, padding=1, deform_groups=1, bias=False).to(DEV)
    pack.conv_offset.weight.data.normal_(0, 0.05)
    pack.conv_offset.bias.data.normal_(0, 0.5)
    x = torch.randn(2, 32, 12, 10, device=DEV)
    want = R.dcn_pack(x.cpu(), pack.conv_offset.weight.cpu(), pack.conv_offset.bias.cpu(), pack.weight.detach().cpu(), None,
                      1, 1, 1, 1)
    assert _rel(pack(x), want) < 2e-4
    net = ResNet(depth=50, out_indices=(2, 3), frozen_stages=1, norm_cfg=dict(type='BN2d', requires_grad=False),
                 norm_eval=True, style='caffe', with_cp=True, dcn=dict(type='DCNv2', deform_groups=1,
                                                                       fallback_on_stride=False),
                 stage_with_dcn=(False, False, True, True), zero_init_residual=False).to(DEV).train()
    neck = FPN(in_channels=[1024, 2048], out_channels=64, num_outs=2).to(DEV)
    for m in net.modules():                                   # non-trivial offsets, or the DCN gradients are untested
        if isinstance(m, ModulatedDeformConv2dPack):
            m.conv_offset.weight.data.normal_(0, 0.01)
    img = torch.randn(1, 2, 3, 64, 96, device=DEV)
    feats = extract_img_feat(img, net, neck)
    assert [tuple(f.shape) for f in feats] == [(1, 2, 64, 4, 6), (1, 2, 64, 2, 3)]
    sum(f.float().square().mean() for f in feats).backward()
    g = net.layer3[0].conv2
    assert g.weight.grad is not None and torch.isfinite(g.weight.grad).all() and float(g.weight.grad.abs().sum()) > 0
    assert float(g.conv_offset.weight.grad.abs().sum()) > 0 and net.conv1.weight.grad is None
