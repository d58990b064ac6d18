"""Host logic of the optimizer's parameter groups (no GPU): the reference's ``paramwise_cfg`` (config :455-462,
``custom_keys = {'img_backbone': dict(lr_mult = 0.1)}``) turned into groups the way mmcv's
DefaultOptimizerConstructor does, and the groups merged into ranges of the flat buffers."""
import torch

from unibev_amd.optim import group_ranges, paramwise_groups


def _named(shapes):
    return [(n, torch.nn.Parameter(torch.zeros(*s))) for n, s in shapes]


def test_custom_keys_scale_lr_and_decay_of_matching_names_only():
    named = _named([('img_backbone.conv1.weight', (4, 3)), ('img_backbone.bn1.bias', (4,)),
                    ('pts_bbox_head.transformer.encoder.weight', (2, 2)), ('pts_bbox_head.bias', (2,))])
    groups = paramwise_groups(named, lr=2e-4, weight_decay=0.01, custom_keys={'img_backbone': dict(lr_mult=0.1)})
    assert [g['lr'] for g in groups] == [2e-4 * 0.1, 2e-4 * 0.1, 2e-4, 2e-4]
    assert [g['weight_decay'] for g in groups] == [0.01] * 4
    assert [g['params'][0] is p for g, (_, p) in zip(groups, named)] == [True] * 4


def test_longest_matching_key_wins_and_frozen_parameters_are_left_out():
    named = _named([('a.b.weight', (1,)), ('a.c.weight', (1,)), ('d.weight', (1,))])
    named[2][1].requires_grad_(False)
    groups = paramwise_groups(named, 1.0, 1.0, custom_keys={'a': dict(lr_mult=0.5), 'a.b': dict(lr_mult=0.25, decay_mult=0.0)})
    assert len(groups) == 2
    assert (groups[0]['lr'], groups[0]['weight_decay']) == (0.25, 0.0)
    assert (groups[1]['lr'], groups[1]['weight_decay']) == (0.5, 1.0)


def test_bias_and_norm_multipliers():
    named = _named([('fc.weight', (2, 2)), ('fc.bias', (2,)), ('norm.weight', (2,)), ('norm.bias', (2,))])
    groups = paramwise_groups(named, 1.0, 0.1, bias_lr_mult=2.0, bias_decay_mult=0.5, norm_decay_mult=0.0,
                              norm_names=('norm',))
    # (mmcv: bias_lr_mult / bias_decay_mult skip a normalisation layer's bias, which takes norm_decay_mult only)
    assert [(g['lr'], g['weight_decay']) for g in groups] == [(1.0, 0.1), (2.0, 0.05), (1.0, 0.0), (1.0, 0.0)]
    import pytest
    with pytest.raises(ValueError, match='dwconv_decay_mult'):
        paramwise_groups(named, 1.0, 0.1, dwconv_decay_mult=0.5)


def test_runs_of_one_setting_merge_into_one_range():
    named = _named([('img_backbone.w', (3, 5)), ('img_backbone.b', (5,)), ('head.w', (7,)), ('head.b', (2,)),
                    ('img_backbone_extra.w', (4,))])
    groups = paramwise_groups(named, 1e-3, 0.01, custom_keys={'img_backbone': dict(lr_mult=0.1)})
    params, ranges = group_ranges(groups, 1e-3, 0.01)
    assert [id(p) for p in params] == [id(p) for _, p in named]
    assert [r[0] for r in ranges] == [20, 29, 33]
    assert [r[1] for r in ranges] == [1e-3 * 0.1, 1e-3, 1e-3 * 0.1]
    # groups without their own lr / weight decay take the defaults
    params, ranges = group_ranges([{'params': [p for _, p in named]}], 5e-4, 0.2)
    assert ranges == [(33, 5e-4, 0.2)]
