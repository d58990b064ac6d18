"""Spatial cross-attention of BEV queries into camera / LiDAR features.

``SpatialCrossAttentionImg`` (reference: models/modules/spatial_cross_attention_img.py:23-215) and
``SpatialCrossAttentionPts`` (models/modules/spatial_cross_attention_pts.py:23-206); same registry
keys, kwargs and state-dict names (``output_proj``, ``deformable_attention.*``).

Camera branch, MI355X form: the reference finds each camera's visible queries with ``nonzero()``
(6 host syncs), zero-pads them to ``max_len`` and runs the sampling op on bs*6 padded batches, then
scatter-adds back.  Here the query Linears run once over the Nq queries (their output does not
depend on the camera) and ``functional.bev_lift`` loops over the cameras a query is visible in,
accumulates in camera order and divides by the per-sample camera count — bit-for-bit the same
sums, no sync, no padding (the reference's quirks q1/q2 are kept: row selection by batch element
0's visibility, count per batch element).
"""
import torch
import torch.nn as nn

from .. import functional as UF
from ..linear import linear as ubv_linear
from ..registry import ATTENTION, build_attention
from .bricks import BaseModule, xavier_init
from .deform_attn import static_hw


@ATTENTION.register_module()
class SpatialCrossAttentionImg(BaseModule):
    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False,
                 deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=256,
                                           num_levels=4),
                 **kwargs):
        super().__init__(init_cfg)
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.init_weight()

    def init_weight(self):
        xavier_init(self.output_proj, distribution='uniform', bias=0.)

    init_weights = init_weight

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, reference_points_cam=None,
                bev_mask=None, level_start_index=None, flag='encoder', **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        if query_pos is not None:
            query = query + query_pos
        bs, num_query, _ = query.size()
        num_cams, l, bs_v, embed_dims = value.shape
        assert num_cams == self.num_cams and bs_v == bs
        value = value.permute(2, 0, 1, 3).reshape(bs * num_cams, l, self.embed_dims)
        da = self.deformable_attention
        if da.can_lift(value) and reference_points_cam.size(3) and \
                da.num_points % reference_points_cam.size(3) == 0:
            hw = static_hw(spatial_shapes)
            assert hw[0][0] * hw[0][1] == l
            vis0 = kwargs.get('cam_vis0')
            count = kwargs.get('cam_count')
            if vis0 is None or count is None:
                seen = bev_mask.any(-1)                                  # (Nc, B, Nq)
                vis0 = seen[:, 0].to(torch.uint8).contiguous()
                count = seen.sum(0).clamp(min=1).to(torch.float32).contiguous()
            if kwargs.get('return_parts') and residual is None and query_pos is None and query.is_cuda:
                # residual through the Linear's pass-through (linear.linear_pass)
                offlog, inp_residual = da.offsets_and_logits(query, passthru=True)
            else:
                offlog = da.offsets_and_logits(query)
            slots = UF.bev_lift(da.project_value_chained(value, kwargs.get('value_chain')), offlog,
                                reference_points_cam, num_cams, hw[0], da.num_heads, da.num_points,
                                vis0=vis0, count=count, query_grid=kwargs.get('query_grid'),
                                visible_lists=kwargs.get('cam_lists') if kwargs.get('cam_vis0') is not None else None)
        else:
            slots = self._masked_path(query, value, reference_points_cam, bev_mask,
                                      spatial_shapes, level_start_index)
        slots = ubv_linear(slots, self.output_proj.weight, self.output_proj.bias)
        if kwargs.get('return_parts'):            # the caller fuses dropout + residual + LayerNorm
            return slots, inp_residual, self.dropout.p
        return self.dropout(slots) + inp_residual

    def _masked_path(self, query, value, reference_points_cam, bev_mask, spatial_shapes,
                     level_start_index):
        """Shapes the fused kernel does not cover (multi-level maps, unusual head sizes): every camera samples
        for EVERY query through the k1 operator, and the rows a camera does not see are zeroed afterwards.
        The reference first gathers each camera's visible rows — chosen by batch element 0's mask, quirk q1 —
        into padded per-camera batches (spatial_cross_attention_img.py:141-212), which needs their counts on the
        host; masking gives the same sums without a read-back, at num_cams times the sampling work."""
        bs, num_query, C = query.shape
        seen = bev_mask.any(-1)                                          # (Nc, B, Nq)
        per_cam = value.view(bs, self.num_cams, -1, C)
        slots = torch.zeros_like(query)
        for cam in range(self.num_cams):
            v = per_cam[:, cam].contiguous()
            out = self.deformable_attention(query=query, key=v, value=v,
                                            reference_points=reference_points_cam[cam],
                                            spatial_shapes=spatial_shapes, level_start_index=level_start_index)
            slots = slots + out * seen[cam, 0].to(out.dtype)[None, :, None]
        count = seen.sum(0).clamp(min=1).to(slots.dtype)                 # (B, Nq): cameras that see the query
        return slots / count[..., None]


@ATTENTION.register_module()
class SpatialCrossAttentionPts(BaseModule):
    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False, deformable_attention=None, **kwargs):
        super().__init__(init_cfg)
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.init_weights()

    def init_weights(self):
        xavier_init(self.output_proj, distribution='uniform', bias=0.)

    def forward(self, query, key, value, residual=None, query_pos=None, spatial_shapes=None,
                reference_points_lidar=None, bev_mask=None, level_start_index=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        if query_pos is not None:
            query = query + query_pos
        bs = query.size(0)
        value = value.permute(1, 0, 2)                               # (hw, bs, C) -> (bs, hw, C)
        reference_points_lidar = reference_points_lidar.permute(1, 2, 0, 3)   # (bs, Nq, Z, 2)
        want_alias = bool(kwargs.get('return_parts')) and residual is None and query_pos is None
        queries = self.deformable_attention(
            query=query, key=value, value=value, reference_points=reference_points_lidar,
            spatial_shapes=spatial_shapes, level_start_index=level_start_index,
            query_grid=kwargs.get('query_grid'), ref_is_grid=kwargs.get('ref_is_grid', False),
            want_query_alias=want_alias, value_chain=kwargs.get('value_chain'))
        if want_alias:
            queries, alias = queries
            if alias is not None:
                inp_residual = alias
        queries = queries.view(bs, -1, self.embed_dims)
        out = ubv_linear(queries, self.output_proj.weight, self.output_proj.bias)
        if kwargs.get('return_parts'):
            return out, inp_residual, self.dropout.p
        return self.dropout(out) + inp_residual
