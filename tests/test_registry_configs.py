"""Host logic of the plugin boundary (CPU only): registry keys, config loading, constructor
semantics, state-dict names, modality-dropout RNG protocol, initialisers, error behaviour."""
import json
import os

import numpy as np
import pytest
import torch

from _util import golden
import make_golden as mg
from unibev_amd import build_attention, build_transformer, configs, load_config
from unibev_amd import registry as reg

REF_CFG = '/root/reference/projects/UniBEV/configs/unibev'
have_ref = pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference configs not present')


def test_registry_keys_of_the_reference_are_present():
    for k in ('SpatialCrossAttentionImg', 'MSDeformableAttention3DImg', 'SpatialCrossAttentionPts',
              'MSDeformableAttention3DPts', 'CustomMSDeformableAttention',
              'MultiScaleDeformableAttention', 'MultiheadAttention',
              'MSDeformableAttention3DUniQueryImg'):
        assert k in reg.ATTENTION, k
    for k in ('ImgEncoder', 'PtsEncoder'):
        assert k in reg.TRANSFORMER_LAYER_SEQUENCE
    for k in ('ImgLayer', 'PtsLayer', 'DetrTransformerDecoderLayer', 'BaseTransformerLayer'):
        assert k in reg.TRANSFORMER_LAYER
    assert 'UniBEVTransformer' in reg.TRANSFORMER
    assert 'UniBEV_Head' in reg.HEADS
    assert 'FFN' in reg.FEEDFORWARD_NETWORK
    assert 'LearnedPositionalEncoding' in reg.POSITIONAL_ENCODING


def test_build_from_cfg_errors_like_mmcv():
    with pytest.raises(KeyError):
        build_attention(dict(type='NoSuchAttention'))
    with pytest.raises(KeyError):
        reg.build_from_cfg(dict(embed_dims=3), reg.ATTENTION)
    with pytest.raises(TypeError):
        reg.build_from_cfg(['x'], reg.ATTENTION)
    with pytest.raises(ValueError):
        build_transformer(configs.transformer_cfg(embed_dims=64, fusion_method='sum', num_layers=1))
    with pytest.raises(ValueError):
        build_attention(dict(type='MSDeformableAttention3DImg', embed_dims=100, num_heads=8))


@have_ref
@pytest.mark.parametrize('fname,kw', [
    ('unibev_nus_LC_cnw_256_modality_dropout.py', dict()),
    ('unibev_nus_LC_cnw_dual_queries_modality_dropout.py', dict(dual_queries=True)),
    ('unibev_nus_LC_avg_256_modality_dropout.py', dict(fusion_method='avg', feature_norm=None)),
    ('unibev_nus_LC_cat_128_modality_dropout.py',
     dict(embed_dims=128, fusion_method='cat', feature_norm=None)),
    ('unibev_nus_L.py', dict(modalities='L', feature_norm=None, drop_modality=None)),
    ('unibev_nus_C.py', dict(modalities='C', feature_norm=None, drop_modality=None,
                             img_da_type='MSDeformableAttention3DUniQueryImg')),
])
def test_shipped_configs_load_and_build_unchanged(fname, kw):
    """The reference's config files are consumed as they are: the transformer sub-tree builds,
    including the C config's unregistered attention name (quirk q10)."""
    cfg = load_config(os.path.join(REF_CFG, fname))
    tcfg = cfg.model.pts_bbox_head.transformer
    ours = configs.transformer_cfg(decoder=dict(tcfg.decoder), **kw)
    def strip(d):        # a key set to None is the same as an absent key for every constructor
        if isinstance(d, dict):
            return {k: strip(v) for k, v in d.items() if v is not None}
        return [strip(v) for v in d] if isinstance(d, list) else d
    a = strip(json.loads(json.dumps(tcfg)))
    b = strip(json.loads(json.dumps(ours)))
    if fname == 'unibev_nus_L.py':
        a.pop('fusion_method', None), b.pop('fusion_method', None)
    assert a == b
    model = build_transformer(tcfg)
    assert model.decoder is not None and model.decoder.num_layers == 6
    if 'img_encoder' in tcfg:
        assert model.img_bev_encoder.num_layers == 3
        layer = model.img_bev_encoder.layers[0]
        assert layer.operation_order == ('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')
        assert layer.attentions[0].num_points == 4 and layer.attentions[0].batch_first
        assert layer.attentions[1].deformable_attention.num_points == 8
        assert layer.ffns[0].feedforward_channels == 2 * tcfg.embed_dims
    head = reg.HEADS.build(cfg.model.pts_bbox_head)
    if kw.get('dual_queries'):            # unibev_head.py:126-133: one query table per modality
        assert model.dual_queries and not hasattr(head, 'bev_embedding')
        assert head.bev_embedding_img.weight.shape == head.bev_embedding_pts.weight.shape == (200 * 200, tcfg.embed_dims)
    else:
        assert head.bev_embedding.weight.shape == (200 * 200, tcfg.embed_dims)


@have_ref
def test_inference_config_base_inheritance():
    cfg = load_config(os.path.join(REF_CFG, 'inference', 'unibev_val_C_full.py'))
    assert cfg.model.pts_bbox_head.transformer.type == 'UniBEVTransformer'
    assert cfg.dist_params.backend == 'gloo'


@pytest.mark.parametrize('name', list(mg.ENCODER_CASES) + ['fullsize'])
def test_state_dict_names_and_shapes_match_the_reference(name):
    g = golden('encoder_' + name)
    cfg = json.loads(str(g['cfg_json']))
    model = build_transformer(cfg)
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith('decoder')}
    ref = {str(n): tuple(json.loads(str(s))) for n, s in zip(g['param_names'], g['param_shapes'])}
    assert ours == ref


def test_encoder_side_parameter_count():
    """Same parameter count as the reference module built from the shipped-size config
    (names/shapes recorded in the full-size fixture)."""
    g = golden('encoder_fullsize')
    ref_n = sum(int(np.prod(json.loads(str(s)))) for s in g['param_shapes'])
    model = build_transformer(configs.transformer_cfg(decoder=None))
    n = sum(p.numel() for k, p in model.named_parameters() if not k.startswith('decoder'))
    assert n == ref_n == 3614659


def test_modality_dropout_draws_match_the_reference():
    """np.random protocol of transformer_fusion.py:463-477: same seed, same dropped modalities."""
    g = golden('modality_dropout')
    for tag, dm in dict(float=0.5, dict=dict(dropout_prob=0.6, lidar_prob=0.3)).items():
        model = build_transformer(configs.transformer_cfg(embed_dims=128, num_layers=1, num_cams=2,
                                                          drop_modality=dm))
        model.train()
        np.random.seed(1234)
        flags = []
        for _ in range(12):
            model._draw_modality_flags([torch.zeros(2, 1)], [torch.zeros(2, 1)])
            flags.append((int(model.c_flag), int(model.l_flag)))
        np.testing.assert_array_equal(np.asarray(flags), g[tag + '_flags'])
    model.eval()
    model._draw_modality_flags([torch.zeros(2, 1)], None)
    assert (model.c_flag, model.l_flag) == (1, 0)
    model.drop_modality = 'half'
    model.train()
    with pytest.raises(ValueError):
        model._draw_modality_flags([torch.zeros(2, 1)], [torch.zeros(2, 1)])


def test_init_weights_follow_the_reference():
    g = golden('init')
    model = build_transformer(configs.transformer_cfg(embed_dims=128, num_layers=1, num_cams=2))
    model.init_weights()
    sd = model.state_dict()
    pre = 'img_bev_encoder.layers.0.attentions.'
    np.testing.assert_allclose(sd[pre + '0.sampling_offsets.bias'].numpy(), g['self_bias'], atol=1e-7)
    np.testing.assert_allclose(sd[pre + '1.deformable_attention.sampling_offsets.bias'].numpy(),
                               g['cross_bias'], atol=1e-7)
    assert sd[pre + '0.sampling_offsets.weight'].abs().sum() == g['self_w_abs'][0] == 0
    assert sd[pre + '0.attention_weights.weight'].abs().sum() == 0
    assert sd[pre + '1.deformable_attention.value_proj.bias'].abs().sum() == 0
    assert sd[pre + '1.output_proj.weight'].abs().sum() > 0


def test_layer_constructor_semantics():
    """BaseTransformerLayer: deprecated kwargs folded into the FFN config, batch_first injected."""
    from unibev_amd.registry import build_transformer_layer
    cfg = configs.transformer_cfg(embed_dims=64, num_layers=1)['img_encoder']['transformerlayers']
    layer = build_transformer_layer(cfg)
    assert layer.batch_first and all(a.batch_first for a in layer.attentions)
    assert layer.ffns[0].layers[0][0].out_features == 128
    assert layer.ffns[0].layers[0][2].p == pytest.approx(0.1)
    assert [type(n).__name__ for n in layer.norms] == ['LayerNorm'] * 3
    assert not layer.pre_norm and layer.num_attn == 2


def test_reference_points_and_positional_encoding_on_cpu():
    """Pure host-side tensor construction (no kernels): identical to the reference's values."""
    from unibev_amd.modules.encoders import ImgEncoder
    from unibev_amd.modules import LearnedPositionalEncoding
    from oracle import unibev_ref as R
    g = golden('point_sampling')
    H, W, D, bs = [int(x) for x in g['small_meta'][:4]]
    np.testing.assert_array_equal(
        ImgEncoder.get_reference_points(H, W, 8, D, dim='3d', bs=bs, device='cpu').numpy(),
        g['small_ref3d'])
    np.testing.assert_array_equal(
        ImgEncoder.get_reference_points(H, W, dim='2d', bs=bs, device='cpu').numpy(), g['small_ref2d'])
    pe = LearnedPositionalEncoding(8, 6, 7)
    pe.init_weights()
    pos = pe(torch.zeros(2, 5, 7))
    ref = R.learned_positional_encoding(pe.row_embed.weight, pe.col_embed.weight, 2, 5, 7)
    assert pos.shape == (2, 16, 5, 7)
    torch.testing.assert_close(pos, ref)
    # the flatten/permute the transformer applies is a free view of the token-major buffer
    assert pos.flatten(2).permute(2, 0, 1)[:, 0].is_contiguous()


@have_ref
@pytest.mark.parametrize('fname,lidar,camera', [
    ('unibev_nus_LC_cnw_256_modality_dropout.py', True, True),
    ('unibev_nus_LC_avg_256_modality_dropout.py', True, True),
    ('unibev_nus_LC_cat_128_modality_dropout.py', True, True),
    ('unibev_nus_L.py', True, False),
    ('unibev_nus_C.py', False, True),
])
def test_shipped_configs_build_the_detector(fname, lidar, camera):
    """``cfg.model`` (type 'UniBEV', unibev_detector.py:17) builds from the reference's config files as they are:
    every sub-module under the attribute name and with the state-dict prefix the published checkpoints use."""
    cfg = load_config(os.path.join(REF_CFG, fname))
    assert cfg.model.type == 'UniBEV' and 'UniBEV' in reg.DETECTORS
    det = reg.DETECTORS.build(cfg.model)
    assert (det.use_lidar, det.use_camera, det.use_radar) == (lidar, camera, False)
    assert det.use_grid_mask and det.grid_mask.prob == 0.7 and det.grid_mask.mode == 1
    assert det.with_pts_bbox and det.pts_bbox_head.transformer.fusion_method == (det.fusion_method or 'linear')
    assert det.with_img_backbone == camera and det.with_img_neck == camera
    assert det.with_pts_backbone == lidar and det.with_pts_neck == lidar
    assert det.with_voxel_encoder == lidar and det.with_middle_encoder == lidar
    keys = set(det.state_dict())
    if camera:
        assert {'img_backbone.layer3.22.conv2.conv_offset.weight', 'img_neck.lateral_convs.0.conv.weight'} <= keys
    if lidar:
        assert det.pts_voxel_layer.max_voxels == (90000, 120000) and det.pts_voxel_layer.max_num_points == 10
        assert det.pts_voxel_encoder.num_features == 5
        assert {'pts_middle_encoder.conv_input.0.weight', 'pts_backbone.blocks.1.15.weight',
                'pts_neck.deblocks.1.0.weight'} <= keys
    assert 'pts_bbox_head.transformer.reference_points.weight' in keys
    assert any(k.startswith('pts_bbox_head.bev_embedding') for k in keys)
    assert det.pts_bbox_head.cls_branches[0][-1].out_features == 10          # FocalLoss: sigmoid classification
    for m in ('extract_img_feat', 'extract_pts_feat', 'extract_feat', 'voxelize', 'forward', 'forward_train',
              'forward_test', 'simple_test', 'forward_dummy', 'forward_outs', 'forward_bev', 'init_weights'):
        assert callable(getattr(det, m)), m
    with pytest.raises(NotImplementedError):
        det.pts_bbox_head.loss(None, None, {})
    with pytest.raises(NotImplementedError):
        det.pts_bbox_head.get_bboxes({}, [])


def test_detector_constructor_errors_and_modality_switch():
    from unibev_amd.modules import UniBEV
    head = configs.head_cfg(embed_dims=32, bev_h=4, bev_w=4, num_query=5, num_layers=1, decoder_layers=1, num_cams=2)
    det = UniBEV(use_lidar=True, use_camera=True, use_radar=True, pts_bbox_head=head)
    with pytest.raises(ValueError, match='Unsupported Modality Mode'):
        det._select_pts_feats([1], [2])
    det = UniBEV(use_lidar=False, use_camera=True, pts_bbox_head=head)
    assert det._select_pts_feats([1], None) is None and not det.with_pts_backbone
    assert det.extract_pts_feat([torch.zeros(1, 5)]) is None and det.extract_img_feat(None) is None
    with pytest.raises(NotImplementedError):
        UniBEV(pts_bbox_head=head, img_roi_head=dict(type='x'))
    with pytest.raises(TypeError):
        det.forward_test(img_metas=dict())
    # softmax classification (mmdet's DETRHead default) has one more output than sigmoid / focal
    soft = dict(head, loss_cls=dict(type='CrossEntropyLoss'))
    assert reg.HEADS.build(soft).cls_out_channels == 11 and reg.HEADS.build(head).cls_out_channels == 10
    assert reg.HEADS.build({k: v for k, v in head.items() if k != 'loss_cls'}).cls_out_channels == 11


def test_fusion_variants_build_with_the_reference_state_dict_names():
    """feature_norm / use_modal_embeds variants no shipped config selects (transformer_fusion.py:136-180): the modules
    build on CPU and expose exactly the parameter names and shapes recorded from the reference's own modules."""
    import json
    import os
    import sys
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, 'golden'))
    import make_golden as mg
    from unibev_amd import build_transformer
    for name in mg.VARIANT_CASES:
        g = np.load(os.path.join(here, 'golden', f'variant_{name}.npz'))
        model = build_transformer(json.loads(str(g['cfg_json'])))
        model.init_weights()
        want = {str(n): tuple(json.loads(str(s))) for n, s in zip(g['param_names'], g['param_shapes'])}
        have = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith('decoder')}
        assert have == want, name
    with pytest.raises(AssertionError):                     # the modality projection is defined for cat fusion only
        build_transformer(configs.transformer_cfg(embed_dims=32, num_layers=1, feature_norm='ModalityProjection'))
