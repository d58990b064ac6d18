"""Numeric parity of the dense backbones / necks against the independent functional
restatement in oracle/backbones_ref.py (published mmdet / mmdet3d forward passes over the checkpoint names; unpinned
by the reference, which does not vendor them).  Same seeded state dict on both sides, forward AND gradients.

CPU part: the plain-convolution variants (host wiring: strides, styles, shortcut rules, FPN level logic, names).
GPU part (``-m gpu``): the shipped-config variants on the device — ResNet with DCNv2 stages through the HIP
deformable convolution, frozen caffe-style norms, FPN 'on_output', SECOND / SECONDFPN through MIOpen."""
import numpy as np
import pytest
import torch

from oracle import backbones_ref as R
from unibev_amd.modules import FPN, SECOND, SECONDFPN, ResNet


def _randomize(module, seed):
    """Seeded, non-degenerate parameters and running statistics (zero-initialised residual norms / offset
    convolutions would hide wiring errors)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, v in module.state_dict().items():
            if name.endswith('num_batches_tracked'):
                continue
            if name.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif name.endswith('running_mean'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif 'conv_offset' in name:
                v.copy_(torch.randn(v.shape, generator=g) * (0.02 if name.endswith('weight') else 0.3))
            elif v.dim() == 1 and ('bn' in name or '.1.' in name or name.endswith('.1.weight')) and name.endswith('weight'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif v.dim() == 1:
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            else:
                fan_in = v[0].numel()
                v.copy_(torch.randn(v.shape, generator=g) * (1.5 / fan_in) ** 0.5)
    return {k: v.detach().clone().double().cpu() for k, v in module.state_dict().items()
            if not k.endswith('num_batches_tracked')}


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # relative L2 distance: a ReLU / max-pool decision that flips under f32 round-off changes single elements by
    # O(1), which a max-norm would report as a wiring error
    err = (a - b).norm().item() / max(b.norm().item(), 1e-30)
    assert err <= tol, f'{what}: relative L2 error {err:.3e} > {tol:.1e}'


def _compare(module, ref_fn, x, tol_f, tol_g, dev='cpu', grad_names=(), seed=0):
    """forward + backward of ``module(x)`` (a tuple / list of maps) against ``ref_fn(P, x64)`` in f64 on the CPU.
    On the CPU the module itself runs in f64 (a wiring check to 1e-9, free of ReLU decisions flipping under f32
    round-off); on the device it runs in f32, the arithmetic the reference uses."""
    P = {k: (v if 'running_' in k else v.requires_grad_()) for k, v in _randomize(module, seed).items()}
    module = module.to(dev) if dev != 'cpu' else module.double()
    xg = x.detach().clone().to(dev).requires_grad_() if dev != 'cpu' else x.detach().double().requires_grad_()
    outs = module(xg)
    x64 = x.detach().double().requires_grad_()
    refs = ref_fn(P, x64)
    assert len(outs) == len(refs)
    rs = np.random.RandomState(seed + 1)
    loss, rloss = 0, 0
    for o, r in zip(outs, refs):
        _close(o, r, tol_f, 'forward')
        cot = torch.from_numpy(rs.standard_normal(tuple(r.shape)))
        loss = loss + (o * cot.to(o)).sum()
        rloss = rloss + (r * cot).sum()
    loss.backward()
    rloss.backward()
    _close(xg.grad, x64.grad, tol_g, 'd(input)')
    params = dict(module.named_parameters())
    for n in grad_names:
        assert params[n].grad is not None, n
        _close(params[n].grad, P[n].grad, tol_g, 'd(' + n + ')')


# ------------------------------------------------------------------------------------------------ CPU: host wiring
@pytest.mark.parametrize('depth,style', [(18, 'pytorch'), (50, 'caffe'), (50, 'pytorch')])
def test_plain_resnet_matches_the_functional_restatement(depth, style):
    torch.manual_seed(0)
    m = ResNet(depth=depth, out_indices=(0, 1, 2, 3), style=style, norm_eval=True, zero_init_residual=False).train()
    x = torch.randn(2, 3, 40, 56)
    last = 'layer4.1.conv2.weight' if depth == 18 else 'layer4.2.conv1.weight'
    _compare(m, lambda P, x: R.resnet(P, x, depth, style=style), x, 1e-9, 1e-9,
             grad_names=('conv1.weight', 'layer2.0.downsample.0.weight', last))


def test_resnet_batch_statistics_mode():
    m = ResNet(depth=18, out_indices=(3,), norm_eval=False).train()
    _compare(m, lambda P, x: R.resnet(P, x, 18, out_indices=(3,), bn_training=True), torch.randn(3, 3, 32, 32),
             1e-9, 1e-8, grad_names=('layer1.0.bn1.weight',))


@pytest.mark.parametrize('kw', [
    dict(in_channels=[24], out_channels=16, num_outs=1, add_extra_convs='on_output', relu_before_extra_convs=True),
    dict(in_channels=[8, 16, 24], out_channels=12, num_outs=5, add_extra_convs='on_output',
         relu_before_extra_convs=True),
    dict(in_channels=[8, 16, 24], out_channels=12, num_outs=5, add_extra_convs='on_input'),
    dict(in_channels=[8, 16, 24], out_channels=12, num_outs=4, add_extra_convs='on_lateral'),
    dict(in_channels=[8, 16, 24], out_channels=12, num_outs=5),
    dict(in_channels=[8, 16, 24], out_channels=12, num_outs=2, start_level=1),
])
def test_fpn_matches_the_functional_restatement(kw):
    m = FPN(**kw)
    P = _randomize(m, 3)
    rs = np.random.RandomState(0)
    ins = [torch.from_numpy(rs.standard_normal((2, c, 20 >> i, 28 >> i)).astype(np.float32)).requires_grad_()
           for i, c in enumerate(kw['in_channels'])]
    outs = m(ins)
    refs = R.fpn(P, [t.detach().double() for t in ins], kw['num_outs'], kw.get('start_level', 0),
                 kw.get('add_extra_convs', False), kw.get('relu_before_extra_convs', False))
    assert len(outs) == len(refs) == kw['num_outs']
    for o, r in zip(outs, refs):
        _close(o, r, 1e-5, 'fpn level')


def test_second_and_secondfpn_match_the_functional_restatement():
    b = SECOND(in_channels=16, out_channels=[8, 16], layer_nums=[2, 3], layer_strides=[1, 2]).train()
    _compare(b, lambda P, x: R.second(P, x, [2, 3], [1, 2]), torch.randn(2, 16, 12, 12), 1e-9, 1e-8,
             grad_names=('blocks.0.0.weight', 'blocks.1.4.weight'))
    for use_conv in (True, False):
        n = SECONDFPN(in_channels=[8, 16], out_channels=[12, 12], upsample_strides=[1, 2],
                      use_conv_for_no_stride=use_conv).train()
        P = _randomize(n, 5)
        xs = [torch.randn(2, 8, 12, 12), torch.randn(2, 16, 6, 6)]
        out = n(xs)
        ref = R.second_fpn(P, [t.double() for t in xs], [1, 2], use_conv)
        assert len(out) == 1
        _close(out[0], ref[0], 1e-4, 'second_fpn')
    n.eval()                                # (the training passes above moved the running statistics)
    P = {k: v.detach().double() for k, v in n.state_dict().items()}
    _close(n(xs)[0], R.second_fpn(P, [t.double() for t in xs], [1, 2], False, bn_training=False)[0], 1e-5, 'eval')


# ------------------------------------------------------------------------------------------------ GPU: shipped variants
@pytest.mark.gpu
def test_resnet_dcn_stages_on_the_device_match_the_oracle():
    """The image backbone of the shipped configs in its structure (caffe style, frozen norms, DCNv2 in stages 3-4,
    checkpointing) at depth 50 and a small image: forward, d(image) and the gradients of a DCN weight, an offset
    convolution and a plain convolution against the f64 restatement.  f32 device GEMMs run as split-bf16 products
    (~2e-6 per product): forward within 2e-4 (relative L2).  Gradients within 8e-2: a pre-activation within f32
    round-off of zero flips its ReLU between the device and the f64 oracle, each flip changes its own gradient path by
    100 % (measured 1.4e-2 on d(image) through 50 layers, 4.1e-2 on an offset-convolution bias whose gradient also
    crosses the bilinear kernel's kinks; bound 8e-2) — a wiring
    error (stride in the wrong convolution, a missing shortcut, swapped offset channels) is O(1)."""
    cfg = dict(depth=50, num_stages=4, out_indices=(2, 3), frozen_stages=1, norm_cfg=dict(type='BN2d', requires_grad=False),
               norm_eval=True, style='caffe', with_cp=True, dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False),
               stage_with_dcn=(False, False, True, True))
    m = ResNet(**cfg).train()
    torch.manual_seed(11)          # (a fixed input: the gradient bound above is a statement about ReLU flips, which differ per input)
    x = torch.randn(2, 3, 96, 160)
    _compare(m, lambda P, x: R.resnet(P, x, 50, out_indices=(2, 3), style='caffe'), x, 2e-4, 8e-2, dev='cuda',
             grad_names=('layer3.1.conv2.weight', 'layer3.1.conv2.conv_offset.weight', 'layer4.0.conv2.conv_offset.bias',
                         'layer2.0.conv2.weight', 'layer4.2.conv3.weight'))


@pytest.mark.gpu
def test_necks_and_lidar_backbone_on_the_device_match_the_oracle():
    from test_backbones_cpu import IMG_NECK, PTS_BACKBONE, PTS_NECK
    from unibev_amd.registry import BACKBONES, NECKS, build_from_cfg
    neck = build_from_cfg(IMG_NECK, NECKS)
    P = _randomize(neck, 7)
    torch.manual_seed(12)
    x = torch.randn(4, 2048, 8, 22)
    out = neck.cuda()([x.cuda()])
    ref = R.fpn(P, [x.double()], 1, 0, 'on_output', True)
    assert len(out) == 1
    _close(out[0], ref[0], 1e-4, 'img_neck')
    b = build_from_cfg(PTS_BACKBONE, BACKBONES).train()
    _compare(b, lambda P, x: R.second(P, x, [5, 5], [1, 2]), torch.randn(2, 256, 36, 36), 2e-4, 3e-2, dev='cuda',
             grad_names=('blocks.0.0.weight', 'blocks.1.15.weight'))
    n = build_from_cfg(PTS_NECK, NECKS).train()
    Pn = _randomize(n, 9)
    xs = [torch.randn(2, 128, 36, 36), torch.randn(2, 256, 18, 18)]
    out = n.cuda()([t.cuda() for t in xs])
    _close(out[0], R.second_fpn(Pn, [t.double() for t in xs], [1, 2], True)[0], 2e-4, 'pts_neck')
