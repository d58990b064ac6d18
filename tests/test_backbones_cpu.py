"""Backbones / necks of SURVEY.md section 8 row f4 (modules/backbones.py): registry keys, the shipped config's kwargs,
state-dict names of the [ext] mmdet / mmdet3d classes (so published checkpoints load) and output shapes — CPU."""
import torch

from unibev_amd.modules import FPN, SECOND, SECONDFPN, ResNet, extract_img_feat
from unibev_amd.registry import BACKBONES, NECKS, build_from_cfg

IMG_BACKBONE = dict(type='ResNet', depth=101, num_stages=4, out_indices=(3,), frozen_stages=1,
                    norm_cfg=dict(type='BN2d', requires_grad=False), norm_eval=True, style='caffe', with_cp=True,
                    dcn=dict(type='DCNv2', deform_groups=1, fallback_on_stride=False),
                    stage_with_dcn=(False, False, True, True))
IMG_NECK = dict(type='FPN', in_channels=[2048], out_channels=256, start_level=0, add_extra_convs='on_output',
                num_outs=1, relu_before_extra_convs=True)
PTS_BACKBONE = dict(type='SECOND', in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2],
                    norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01), conv_cfg=dict(type='Conv2d', bias=False))
PTS_NECK = dict(type='SECONDFPN', in_channels=[128, 256], upsample_strides=[1, 2], out_channels=[128, 128],
                norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01), upsample_cfg=dict(type='deconv', bias=False),
                use_conv_for_no_stride=True)


def test_config_dicts_build_and_carry_the_checkpoint_names():
    r = build_from_cfg(IMG_BACKBONE, BACKBONES)
    sd = r.state_dict()
    for k in ('conv1.weight', 'bn1.running_mean', 'layer1.0.downsample.0.weight', 'layer3.22.conv2.conv_offset.weight',
              'layer3.22.conv2.weight', 'layer4.2.conv2.conv_offset.bias', 'layer4.2.bn3.weight'):
        assert k in sd, k
    assert 'layer2.0.conv2.conv_offset.weight' not in sd                     # stages 1-2 are plain convolutions
    assert sd['layer3.0.conv2.conv_offset.weight'].shape == (27, 256, 3, 3) and sd['layer4.0.conv2.weight'].shape == (512, 512, 3, 3)
    assert sd['layer3.0.conv1.weight'].shape == (256, 512, 1, 1)
    assert sum(v.numel() for k, v in sd.items() if 'num_batches' not in k) > 44e6
    # frozen: the stem and stage 1, every batch norm; DCN stages stay trainable
    assert not r.conv1.weight.requires_grad and not r.layer1[0].conv1.weight.requires_grad
    assert r.layer3[0].conv2.weight.requires_grad and not r.layer3[0].bn2.weight.requires_grad
    r.train()
    assert not r.layer3[0].bn2.training and not r.bn1.training                # norm_eval
    # caffe style: the stride of a stage sits in the first 1x1 convolution
    assert r.layer2[0].conv1.stride == (2, 2) and r.layer2[0].conv2.stride == (1, 1)
    n = build_from_cfg(IMG_NECK, NECKS)
    assert list(n.state_dict()) == ['lateral_convs.0.conv.weight', 'lateral_convs.0.conv.bias',
                                    'fpn_convs.0.conv.weight', 'fpn_convs.0.conv.bias']


def test_lidar_backbone_and_neck_shapes_and_names():
    b, n = build_from_cfg(PTS_BACKBONE, BACKBONES), build_from_cfg(PTS_NECK, NECKS)
    assert isinstance(b, SECOND) and isinstance(n, SECONDFPN)
    x = torch.randn(2, 256, 36, 36)
    feats = b(x)
    assert [tuple(f.shape) for f in feats] == [(2, 128, 36, 36), (2, 256, 18, 18)]
    out = n(feats)
    assert len(out) == 1 and out[0].shape == (2, 256, 36, 36)
    assert 'blocks.1.15.weight' in b.state_dict() and b.blocks[0][0].bias is None
    sd = n.state_dict()
    assert sd['deblocks.0.0.weight'].shape == (128, 128, 1, 1)               # stride 1 + use_conv_for_no_stride: a 1x1 conv
    assert sd['deblocks.1.0.weight'].shape == (256, 128, 2, 2)               # transposed convolution: [in, out, k, k]
    assert b.blocks[0][1].eps == 1e-3 and b.blocks[0][1].momentum == 0.01


def test_plain_resnet_and_fpn_forward_on_cpu():
    r = ResNet(depth=18, out_indices=(1, 2, 3), norm_eval=False)
    f = FPN(in_channels=[128, 256, 512], out_channels=32, num_outs=4, add_extra_convs='on_output',
            relu_before_extra_convs=True)
    img = torch.randn(1, 2, 3, 64, 96)
    feats = extract_img_feat(img, r, f)
    assert [tuple(t.shape) for t in feats] == [(1, 2, 32, 8, 12), (1, 2, 32, 4, 6), (1, 2, 32, 2, 3), (1, 2, 32, 1, 2)]
    g = FPN(in_channels=[128, 256, 512], out_channels=16, num_outs=5)        # no extra convs: strided max-pool levels
    assert len(g(r(img[0]))) == 5 and len(g.fpn_convs) == 3
