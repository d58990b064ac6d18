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
