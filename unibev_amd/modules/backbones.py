"""Dense backbones and necks of the two branches under the reference's registry keys,
constructor kwargs and state-dict names, so the shipped configs build and published checkpoints load:

  img_backbone  ``ResNet`` (depth 101, caffe style, frozen BN, DCNv2 in stages 3-4)   config :225-236
  img_neck      ``FPN`` (2048 -> 256, one output level)                                 config :237-244
  pts_backbone  ``SECOND``                                                              config :209-216
  pts_neck      ``SECONDFPN``                                                           config :217-224

Reference classes: [ext] mmdet 2.14 ``models/backbones/resnet.py`` / ``necks/fpn.py``, mmdet3d 0.17
``models/backbones/second.py`` / ``necks/second_fpn.py``, mmcv ``ModulatedDeformConv2dPack``.  The plain
convolutions, batch norms and transposed convolutions are MIOpen calls through torch.nn (the survey's plan for this
row); the one custom operator, the modulated deformable convolution, is ``functional.modulated_deform_conv2d``
(``csrc/deform_conv.hip`` + the MFMA GEMMs).  ``extract_img_feat`` mirrors
``UniBEV.extract_img_feat`` (models/detectors/unibev_detector.py:86-110).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils import checkpoint as cp

from .. import functional as UF
from ..registry import BACKBONES, NECKS


def _norm(cfg, channels):
    cfg = dict(cfg or dict(type='BN'))
    kind = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    if kind not in ('BN', 'BN2d', 'naiveSyncBN2d', 'SyncBN'):
        raise NotImplementedError(f'norm layer {kind}')
    bn = nn.BatchNorm2d(channels, **cfg)
    for p in bn.parameters():
        p.requires_grad = requires_grad
    return bn


class ModulatedDeformConv2dPack(nn.Module):
    """[ext] mmcv ``ModulatedDeformConv2dPack`` (registry names 'DCNv2'): ``conv_offset`` predicts 2 offsets + 1 mask
    logit per tap and deformable group (zero-initialised: the layer starts as 0.5 x a plain convolution)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=True, **kwargs):
        super().__init__()
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = pair(kernel_size), pair(stride), pair(padding), pair(dilation)
        self.groups, self.deform_groups = groups, deform_groups
        if groups != 1:
            raise NotImplementedError('ModulatedDeformConv2dPack: groups != 1 is not built')
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     self.kernel_size, self.stride, self.padding, self.dilation, bias=True)
        self.init_weights()

    def init_weights(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return UF.modulated_deform_conv2d(x, offset, torch.sigmoid(mask), self.weight, self.bias, self.stride,
                                          self.padding, self.dilation, self.groups, self.deform_groups)


def _conv3x3(cin, cout, stride, dilation, dcn):
    if dcn is not None:
        cfg = dict(dcn)
        kind = cfg.pop('type')
        cfg.pop('fallback_on_stride', None)
        if kind != 'DCNv2':
            raise NotImplementedError(f'dcn type {kind}')
        return ModulatedDeformConv2dPack(cin, cout, 3, stride=stride, padding=dilation, dilation=dilation, bias=False,
                                         **cfg)
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 norm_cfg=None, dcn=None):
        super().__init__()
        assert dcn is None, 'DCN in BasicBlock is not built (mmdet has none either)'
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn1 = _norm(norm_cfg, planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _norm(norm_cfg, planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.with_cp = downsample, with_cp

    def _inner(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        return self.bn2(self.conv2(out)) + identity

    def forward(self, x):
        out = cp.checkpoint(self._inner, x, use_reentrant=False) if self.with_cp and x.requires_grad else self._inner(x)
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 norm_cfg=None, dcn=None):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)      # caffe: the stride sits in the first 1x1
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = _norm(norm_cfg, planes)
        self.conv2 = _conv3x3(planes, planes, s2, dilation, dcn)
        self.bn2 = _norm(norm_cfg, planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _norm(norm_cfg, planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.with_cp = downsample, with_cp

    def _inner(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.bn3(self.conv3(out)) + identity

    def forward(self, x):
        out = cp.checkpoint(self._inner, x, use_reentrant=False) if self.with_cp and x.requires_grad else self._inner(x)
        return self.relu(out)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
                     101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True,
                 pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if deep_stem or avg_down or plugins is not None or conv_cfg is not None:
            raise NotImplementedError('ResNet: deep_stem / avg_down / plugins / conv_cfg are not built')
        block, stage_blocks = self.arch_settings[depth]
        stem_channels = stem_channels or base_channels
        self.out_indices, self.frozen_stages, self.norm_eval = tuple(out_indices), frozen_stages, norm_eval
        self.zero_init_residual = zero_init_residual
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, 2, 3, bias=False)
        self.bn1 = _norm(norm_cfg, stem_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.res_layers = []
        inplanes = stem_channels
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            stage_dcn = dcn if stage_with_dcn[i] else None
            layers = []
            for j in range(stage_blocks[i]):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                                         _norm(norm_cfg, planes * block.expansion))
                layers.append(block(inplanes, planes, stride, dilations[i], down, style, with_cp, norm_cfg, stage_dcn))
                inplanes = planes * block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self.init_weights()
        self._freeze_stages()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for m in self.modules():
            if isinstance(m, ModulatedDeformConv2dPack):
                m.conv_offset.weight.data.zero_()
                m.conv_offset.bias.data.zero_()
            if self.zero_init_residual and isinstance(m, Bottleneck):
                nn.init.zeros_(m.bn3.weight)
            elif self.zero_init_residual and isinstance(m, BasicBlock):
                nn.init.zeros_(m.bn2.weight)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


class _ConvModule(nn.Module):
    """The slice of mmcv's ConvModule these necks use: conv (+ bias when there is no norm) under the name ``conv``."""

    def __init__(self, cin, cout, k, padding=0, norm_cfg=None, act=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=norm_cfg is None)
        self.bn = _norm(norm_cfg, cout) if norm_cfg is not None else None
        self.act = act

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x) if self.act else x


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        if conv_cfg is not None or act_cfg is not None:
            raise NotImplementedError('FPN: conv_cfg / act_cfg are not built')
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.num_ins = len(in_channels)
        self.relu_before_extra_convs = relu_before_extra_convs
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= self.num_ins and num_outs == end_level - start_level
        self.start_level = start_level
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            add_extra_convs = 'on_input'
        self.add_extra_convs = add_extra_convs
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(_ConvModule(in_channels[i], out_channels, 1,
                                                  norm_cfg=None if no_norm_on_lateral else norm_cfg))
            self.fpn_convs.append(_ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg))
        extra = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = self.in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == 'on_input') \
                    else out_channels
                conv = _ConvModule(cin, out_channels, 3, padding=1, norm_cfg=norm_cfg)
                conv.conv.stride = (2, 2)
                self.fpn_convs.append(conv)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            if 'scale_factor' in self.upsample_cfg:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], **self.upsample_cfg)
            else:
                laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                                   **self.upsample_cfg)
        used = len(laterals)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(used)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - used):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                if self.add_extra_convs == 'on_input':
                    src = inputs[self.backbone_end_level - 1]
                elif self.add_extra_convs == 'on_lateral':
                    src = laterals[-1]
                else:
                    src = outs[-1]
                outs.append(self.fpn_convs[used](src))
                for i in range(used + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)


@BACKBONES.register_module()
class SECOND(nn.Module):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01), conv_cfg=dict(type='Conv2d', bias=False),
                 init_cfg=None, pretrained=None):
        super().__init__()
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        bias = dict(conv_cfg).get('bias', False)
        filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            layers = [nn.Conv2d(filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1, bias=bias),
                      _norm(norm_cfg, out_channels[i]), nn.ReLU(inplace=True)]
            for _ in range(n):
                layers += [nn.Conv2d(out_channels[i], out_channels[i], 3, padding=1, bias=bias),
                           _norm(norm_cfg, out_channels[i]), nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*layers))
        self.blocks = nn.ModuleList(blocks)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        outs = []
        for blk in self.blocks:
            x = blk(x)
            outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01), upsample_cfg=dict(type='deconv', bias=False),
                 conv_cfg=dict(type='Conv2d', bias=False), use_conv_for_no_stride=False, init_cfg=None):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        deblocks = []
        for i, oc in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                up = nn.ConvTranspose2d(in_channels[i], oc, stride, stride=stride,
                                        bias=dict(upsample_cfg).get('bias', False))
            else:
                s = int(round(1 / stride))
                up = nn.Conv2d(in_channels[i], oc, s, stride=s, bias=dict(conv_cfg).get('bias', False))
            deblocks.append(nn.Sequential(up, _norm(norm_cfg, oc), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


def extract_img_feat(img, img_backbone, img_neck=None, grid_mask=None):
    """``UniBEV.extract_img_feat`` (unibev_detector.py:86-110): (B, N, 3, H, W) images -> list of (B, N, C, h, w)."""
    if img is None:
        return None
    B = img.size(0)
    if img.dim() == 5:
        Bn, N, C, H, W = img.size()
        img = img.reshape(Bn * N, C, H, W)
    if grid_mask is not None:
        img = grid_mask(img)
    feats = img_backbone(img)
    if isinstance(feats, dict):
        feats = list(feats.values())
    if img_neck is not None:
        feats = img_neck(feats)
    out = []
    for f in feats:
        BN, C, H, W = f.size()
        out.append(f.view(B, BN // B, C, H, W))
    return out
