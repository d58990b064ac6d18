"""Config dictionaries in the reference's registry-key + kwargs vocabulary.

``transformer_cfg`` produces the ``model.pts_bbox_head.transformer`` sub-tree of the reference's
configs (projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:255-349) for a
chosen size; ``tests/test_registry_configs.py`` checks that, at the reference's own size, it equals
the sub-tree of the shipped config files (decoder excluded: out of scope).
"""
import copy

PC_RANGE = [-54, -54, -5, 54, 54, 3]


def _layer(kind, dim, num_levels, num_points, ffn_dim, da_type=None, num_cams=None):
    cross = 'SpatialCrossAttentionImg' if kind == 'img' else 'SpatialCrossAttentionPts'
    da = da_type or ('MSDeformableAttention3DImg' if kind == 'img' else 'MSDeformableAttention3DPts')
    extra = dict(num_cams=num_cams) if (num_cams is not None and kind == 'img') else {}
    return dict(
        type='ImgLayer' if kind == 'img' else 'PtsLayer',
        attn_cfgs=[
            dict(type='MultiScaleDeformableAttention', embed_dims=dim, num_levels=1),
            dict(type=cross, pc_range=PC_RANGE,
                 deformable_attention=dict(type=da, embed_dims=dim, num_points=num_points,
                                           num_levels=num_levels),
                 embed_dims=dim, **extra),
        ],
        ffn_cfgs=dict(type='FFN', embed_dims=dim),
        feedforward_channels=ffn_dim,
        ffn_dropout=0.1,
        operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'))


def transformer_cfg(embed_dims=256, fusion_method='linear', feature_norm='ChannelNormWeights',
                    drop_modality=0.5, num_layers=3, num_levels=1, num_points=8,
                    pillar_cam=4, pillar_lidar=4, modalities='LC', decoder=None,
                    img_da_type=None, num_cams=None, **extra):
    cfg = dict(type='UniBEVTransformer', embed_dims=embed_dims, fusion_method=fusion_method)
    if drop_modality is not None:
        cfg['drop_modality'] = drop_modality
    if feature_norm is not None:
        cfg['feature_norm'] = feature_norm
    if 'C' in modalities:
        cfg['img_encoder'] = dict(
            type='ImgEncoder', num_layers=num_layers, pc_range=PC_RANGE,
            num_points_in_pillar=pillar_cam, return_intermediate=False,
            transformerlayers=_layer('img', embed_dims, num_levels, num_points, embed_dims * 2,
                                     img_da_type, num_cams))
    if 'L' in modalities:
        cfg['pts_encoder'] = dict(
            type='PtsEncoder', num_layers=num_layers, pc_range=PC_RANGE,
            num_points_in_pillar_lidar=pillar_lidar, return_intermediate=False,
            transformerlayers=_layer('pts', embed_dims, num_levels, num_points, embed_dims * 2))
    cfg['decoder'] = copy.deepcopy(decoder) if decoder is not None else dict(type='NullDecoder')
    if num_cams is not None:
        cfg['num_cams'] = num_cams
    cfg.update(extra)
    return cfg


def decoder_cfg(embed_dims=256, num_layers=6, scale_factor=1):
    """``transformer.decoder`` of the shipped configs
    (configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:325-349): ``scale_factor`` is 2 for
    the cat fusion (``dec_scale_factor``)."""
    dim = embed_dims * scale_factor
    return dict(
        type='DetectionTransformerDecoder', num_layers=num_layers, return_intermediate=True,
        transformerlayers=dict(
            type='DetrTransformerDecoderLayer',
            attn_cfgs=[dict(type='MultiheadAttention', embed_dims=dim, num_heads=8, dropout=0.1),
                       dict(type='CustomMSDeformableAttention', embed_dims=dim, num_levels=1)],
            ffn_cfgs=dict(type='FFN', embed_dims=dim),
            feedforward_channels=embed_dims * 2 * scale_factor, ffn_dropout=0.1,
            operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))


def head_cfg(embed_dims=256, bev_h=200, bev_w=200, num_query=900, num_classes=10,
             decoder_layers=6, with_box_refine=True, **transformer_kw):
    """``model.pts_bbox_head`` of the shipped configs (:245-378) without the loss / assigner
    entries the forward never reads."""
    s = 2 if transformer_kw.get('fusion_method') == 'cat' else 1
    tcfg = transformer_cfg(embed_dims=embed_dims,
                           decoder=decoder_cfg(embed_dims, decoder_layers, s), **transformer_kw)
    return dict(type='UniBEV_Head', bev_h=bev_h, bev_w=bev_w, num_query=num_query,
                num_classes=num_classes, in_channels=embed_dims, sync_cls_avg_factor=True,
                with_box_refine=with_box_refine, as_two_stage=False, transformer=tcfg,
                bbox_coder=dict(type='NMSFreeCoder', post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                pc_range=PC_RANGE, max_num=300, num_classes=num_classes),
                positional_encoding=dict(type='LearnedPositionalEncoding', num_feats=embed_dims // 2,
                                         row_num_embed=bev_h, col_num_embed=bev_w),
                loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0))
