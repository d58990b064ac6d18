"""CPU oracle of the dense backbones and necks — TEST INFRASTRUCTURE ONLY (imported by
tests/ and nothing else).

PARITY UNPINNED BY THE REFERENCE: ``ResNet`` / ``FPN`` are [ext] mmdet 2.19.0 (models/backbones/resnet.py,
models/necks/fpn.py), ``SECOND`` / ``SECONDFPN`` [ext] mmdet3d 0.18.1 (models/backbones/second.py,
models/necks/second_fpn.py); none of them is vendored in /root/reference, which reaches them only through its configs
(configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:209-244).  The restatement is written in an independent
form — pure functions over a state dict keyed by the PUBLISHED checkpoint names, ``torch.nn.functional`` calls only,
no module classes — following the published forward passes:

* ResNet: 7x7/2 stem conv -> BN -> ReLU -> 3x3/2 max-pool (padding 1); stage i has ``blocks[i]`` residual blocks,
  the first with stride ``strides[i]`` and a 1x1 conv + BN shortcut when the shape changes.  Bottleneck = 1x1 -> 3x3 ->
  1x1 (x4 channels); style 'caffe' puts the stride in the FIRST 1x1, 'pytorch' in the 3x3.  BasicBlock = 3x3 -> 3x3.
  With ``dcn`` the 3x3 of a Bottleneck is a ``ModulatedDeformConv2dPack`` (oracle/dcn_ref.py).
* FPN: 1x1 laterals, top-down nearest upsampling added in place, 3x3 output convs, extra levels by stride-2 3x3 convs
  on the input / lateral / output (ReLU in between when ``relu_before_extra_convs``) or by stride-2 max-pool of size 1.
* SECOND: per stage [3x3 conv (stride) -> BN -> ReLU] + n x [3x3 conv -> BN -> ReLU]; outputs of every stage.
* SECONDFPN: per input a transposed conv (kernel = stride) — or, for stride 1 with ``use_conv_for_no_stride``, a 1x1
  conv — then BN -> ReLU; outputs concatenated on channels.
Batch norm: ``training=True`` uses batch statistics (biased variance), else the running statistics.
"""
import torch
import torch.nn.functional as F

from . import dcn_ref

_ARCH = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
         101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}


def _bn(P, name, x, training=False, eps=1e-5):
    if training:
        return F.batch_norm(x, None, None, P[name + '.weight'], P[name + '.bias'], True, 0.0, eps)
    return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'], P[name + '.weight'],
                        P[name + '.bias'], False, 0.0, eps)


def _conv3x3(P, name, x, stride, dilation):
    if name + '.conv_offset.weight' in P:
        return dcn_ref.dcn_pack(x, P[name + '.conv_offset.weight'], P[name + '.conv_offset.bias'], P[name + '.weight'],
                                P.get(name + '.bias'), stride, dilation, dilation, 1)
    return F.conv2d(x, P[name + '.weight'], None, stride, dilation, dilation)


def resnet(P, x, depth, out_indices=(0, 1, 2, 3), strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), style='pytorch',
           num_stages=4, bn_training=False):
    """``P``: state dict of the backbone (names ``conv1.weight``, ``layer3.0.conv2.conv_offset.weight`` ...)."""
    kind, blocks = _ARCH[depth]
    x = F.conv2d(x, P['conv1.weight'], None, 2, 3)
    x = F.max_pool2d(torch.relu(_bn(P, 'bn1', x, bn_training)), 3, 2, 1)
    outs = []
    for i in range(num_stages):
        for j in range(blocks[i]):
            pre = f'layer{i + 1}.{j}'
            s = strides[i] if j == 0 else 1
            idt = x
            if pre + '.downsample.0.weight' in P:
                idt = _bn(P, pre + '.downsample.1', F.conv2d(x, P[pre + '.downsample.0.weight'], None, s),
                          bn_training)
            if kind == 'bottleneck':
                s1, s2 = (s, 1) if style == 'caffe' else (1, s)
                y = torch.relu(_bn(P, pre + '.bn1', F.conv2d(x, P[pre + '.conv1.weight'], None, s1), bn_training))
                y = torch.relu(_bn(P, pre + '.bn2', _conv3x3(P, pre + '.conv2', y, s2, dilations[i]), bn_training))
                y = _bn(P, pre + '.bn3', F.conv2d(y, P[pre + '.conv3.weight']), bn_training)
            else:
                y = torch.relu(_bn(P, pre + '.bn1', F.conv2d(x, P[pre + '.conv1.weight'], None, s, dilations[i],
                                                             dilations[i]), bn_training))
                y = _bn(P, pre + '.bn2', F.conv2d(y, P[pre + '.conv2.weight'], None, 1, 1), bn_training)
            x = torch.relu(y + idt)
        if i in out_indices:
            outs.append(x)
    return outs


def fpn(P, inputs, num_outs, start_level=0, add_extra_convs=False, relu_before_extra_convs=False):
    """``P``: ``lateral_convs.i.conv.{weight,bias}``, ``fpn_convs.i.conv.{weight,bias}`` (no norm, end_level = -1)."""
    if add_extra_convs is True:
        add_extra_convs = 'on_input'
    n_lat = len(inputs) - start_level
    lat = [F.conv2d(inputs[i + start_level], P[f'lateral_convs.{i}.conv.weight'], P[f'lateral_convs.{i}.conv.bias'])
           for i in range(n_lat)]
    for i in range(n_lat - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    outs = [F.conv2d(lat[i], P[f'fpn_convs.{i}.conv.weight'], P[f'fpn_convs.{i}.conv.bias'], 1, 1)
            for i in range(n_lat)]
    while len(outs) < num_outs:
        k = len(outs)
        if not add_extra_convs:
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            continue
        if k == n_lat:
            src = {'on_input': inputs[-1], 'on_lateral': lat[-1], 'on_output': outs[-1]}[add_extra_convs]
        else:
            src = torch.relu(outs[-1]) if relu_before_extra_convs else outs[-1]
        outs.append(F.conv2d(src, P[f'fpn_convs.{k}.conv.weight'], P[f'fpn_convs.{k}.conv.bias'], 2, 1))
    return outs


def second(P, x, layer_nums, layer_strides, eps=1e-3, bn_training=True):
    """``P``: ``blocks.i.{0,3,6,...}.weight`` convolutions, ``blocks.i.{1,4,...}`` batch norms."""
    outs = []
    for i, n in enumerate(layer_nums):
        for j in range(n + 1):
            x = F.conv2d(x, P[f'blocks.{i}.{3 * j}.weight'], P.get(f'blocks.{i}.{3 * j}.bias'),
                         layer_strides[i] if j == 0 else 1, 1)
            x = torch.relu(_bn(P, f'blocks.{i}.{3 * j + 1}', x, bn_training, eps))
        outs.append(x)
    return outs


def second_fpn(P, xs, upsample_strides, use_conv_for_no_stride=False, eps=1e-3, bn_training=True):
    """``P``: ``deblocks.i.0.weight`` ([in, out, k, k] for the transposed convolutions), ``deblocks.i.1`` norms."""
    ups = []
    for i, (x, s) in enumerate(zip(xs, upsample_strides)):
        w = P[f'deblocks.{i}.0.weight']
        if s > 1 or (s == 1 and not use_conv_for_no_stride):
            y = F.conv_transpose2d(x, w, P.get(f'deblocks.{i}.0.bias'), stride=s)
        else:
            k = int(round(1 / s))
            y = F.conv2d(x, w, P.get(f'deblocks.{i}.0.bias'), stride=k)
        ups.append(torch.relu(_bn(P, f'deblocks.{i}.1', y, bn_training, eps)))
    return [torch.cat(ups, 1) if len(ups) > 1 else ups[0]]
