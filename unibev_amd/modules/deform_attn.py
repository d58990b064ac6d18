"""Deformable-attention modules of the BEV encoder, on the HIP kernels.

Registry keys / constructor kwargs / state-dict names follow the reference:
  * ``MultiScaleDeformableAttention`` — [ext] mmcv module in the self-attn slot of every encoder
    layer (configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:270-273); the reference
    vendors a verbatim copy as ``CustomMSDeformableAttention`` (models/modules/decoder.py:131-338).
  * ``MSDeformableAttention3DImg`` / ``MSDeformableAttention3DPts`` —
    models/modules/spatial_cross_attention_img.py:218-442, spatial_cross_attention_pts.py:209-449.
  * alias ``MSDeformableAttention3DUniQueryImg`` (named by unibev_nus_C.py:206, registered nowhere
    in the reference — quirk q10).

Fast path (one level, 4 or 8 points): the two query Linears run as ONE GEMM whose rows
``[offsets | logits]`` feed ``functional.bev_lift`` directly; sampling locations and softmaxed
weights are never materialised.  Anything else composes the k1 operator exactly as the
reference does.
"""
import math
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as UF
from ..linear import linear as ubv_linear
from ..linear import linear_cat, linear_cat_pass, linear_pass, self_attn_in, self_attn_in_supported
from ..registry import ATTENTION
from .bricks import BaseModule, constant_init, xavier_init


def _OFFLOG_F32():
    return os.environ.get('UBV_OFFLOG', '') == 'fp32'


_VALUE_STORAGE = [{'fp16': torch.float16, 'bf16': torch.bfloat16}.get(os.environ.get('UBV_VALUE16', ''))]


def set_value_storage(dtype):
    """VALUE-ONLY 16-bit mode ("fp16/bf16 value for the build"): the residual stream,
    every Linear (split-bf16 MFMA products), offsets, logits, softmax and sampling locations stay f32; only the
    projected value maps the sampling kernels gather from — and with them the sampled output and, in backward,
    ``grad_output`` / ``grad_value`` — are stored in ``dtype`` (torch.float16 / torch.bfloat16; None = off).  Halves
    the gathered bytes.  Returns the previous setting."""
    if dtype not in (None, torch.float16, torch.bfloat16):
        raise ValueError('value storage must be None, torch.float16 or torch.bfloat16')
    prev, _VALUE_STORAGE[0] = _VALUE_STORAGE[0], dtype
    return prev


def _store_value(value):
    st = _VALUE_STORAGE[0]
    if st is not None and value.is_cuda and value.dtype == torch.float32:
        return value.to(st)
    return value


def static_hw(spatial_shapes):
    """Host copy [(h, w), ...] of a ``spatial_shapes`` tensor without a device sync when the
    producer attached one (``_ubv_hw``); otherwise one ``tolist()`` (sync) as in any eager use."""
    hw = getattr(spatial_shapes, '_ubv_hw', None)
    if hw is None:
        hw = [tuple(int(v) for v in r) for r in spatial_shapes.tolist()]
        try:
            spatial_shapes._ubv_hw = hw
        except Exception:
            pass
    return hw


_SHAPE_TENSORS = {}


def shapes_tensor(hw, device, query_grid=None):
    """(L, 2) int64 device tensor carrying its own host copy.  Cached per (shapes, device): the
    upload happens once, so a later forward pass issues no host-to-device copy (none is allowed
    while a HIP graph is being captured).  ``query_grid`` = (qh, qw): the queries the operator will be called with are
    that grid in row-major order (BEV queries) — ``functional.ms_deform_attn`` then takes the TILE plan
    (``ubv_ms_deform_attn_forward_grid``)."""
    key = (tuple(tuple(int(v) for v in r) for r in hw), str(device), None if query_grid is None else tuple(query_grid))
    t = _SHAPE_TENSORS.get(key)
    if t is None:
        t = torch.as_tensor(hw, dtype=torch.long, device=device)
        t._ubv_hw = [tuple(int(v) for v in r) for r in hw]
        if query_grid is not None:
            t._ubv_qgrid = (int(query_grid[0]), int(query_grid[1]))
        if len(_SHAPE_TENSORS) > 64:
            _SHAPE_TENSORS.clear()
        _SHAPE_TENSORS[key] = t
    return t


def index_tensor(values, device):
    """Cached int64 device tensor of a small host list (level start indices)."""
    key = ('idx', tuple(int(v) for v in values), str(device))
    t = _SHAPE_TENSORS.get(key)
    if t is None:
        t = torch.as_tensor([int(v) for v in values], dtype=torch.long, device=device)
        _SHAPE_TENSORS[key] = t
    return t


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError('invalid input for _is_power_of_2: {} (type: {})'.format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class _DeformAttnBase(BaseModule):
    """Parameters and initialisation shared by every deformable attention on the path."""

    def __init__(self, embed_dims, num_heads, num_levels, num_points, im2col_step, batch_first,
                 norm_cfg, init_cfg, with_output_proj):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        if not _is_power_of_2(embed_dims // num_heads):
            warnings.warn("You'd better set embed_dims in MultiScaleDeformAttention to make the "
                          'dimension of each attention head a power of 2 which is more efficient '
                          'in our implementation.')
        self.norm_cfg = norm_cfg
        self.batch_first = batch_first
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims) if with_output_proj else None
        self.init_weights()

    def init_weights(self):
        """Offsets start as the 8 compass directions scaled by the point index, attention logits
        at zero (spatial_cross_attention_img.py:293-311, decoder.py:208-226)."""
        constant_init(self.sampling_offsets, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid_init.view(-1).to(self.sampling_offsets.bias.device)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution='uniform', bias=0.)
        xavier_init(self.output_proj, distribution='uniform', bias=0.)
        self._is_init = True

    init_weight = init_weights

    # -- pieces --------------------------------------------------------------------------------
    def offsets_and_logits(self, query, passthru=False, row_bias=None):
        """One GEMM for both query Linears: rows [H*L*P*2 offsets | H*L*P logits].
        ``passthru``: also return the alias of ``query`` for the caller's residual branch
        (``linear.linear_pass``).  ``row_bias`` (Nq, H*L*P*3): the positional term of these Linears, the same for
        every sample, added in the GEMM epilogue (``encoders._EncoderBase._fold_pos_terms``)."""
        if row_bias is not None:
            assert passthru
            return linear_cat_pass(query, self._ol_weights(),
                                   (self.sampling_offsets.bias, self.attention_weights.bias), row_bias=row_bias)
        if _OFFLOG_F32() and query.is_cuda and torch.is_autocast_enabled('cuda'):
            # offsets / logits computed and kept in f32 under autocast (precision knob: a 16-bit
            # pixel offset of 8 px carries 4e-3 px of rounding)
            with torch.autocast('cuda', enabled=False):
                w = torch.cat((self.sampling_offsets.weight, self.attention_weights.weight), 0)
                b = torch.cat((self.sampling_offsets.bias, self.attention_weights.bias), 0)
                out = F.linear(query.float(), w.float(), b.float())
            return (out, query) if passthru else out
        fn = linear_cat_pass if passthru else linear_cat
        return fn(query, (self.sampling_offsets.weight, self.attention_weights.weight),
                  (self.sampling_offsets.bias, self.attention_weights.bias))

    def _ol_weights(self):
        """(sampling_offsets.weight, attention_weights.weight) for this pass' own GEMM: the aliases the encoder's
        positional fold left for it (``encoders._EncoderBase._fold_pos_terms``: the two consumers of these weights then
        meet in ``functional.fan_out``'s own add instead of the autograd engine's), else the parameters.  Single use."""
        al = self.__dict__.pop('_ubv_w_alias', None)
        return al if al is not None else (self.sampling_offsets.weight, self.attention_weights.weight)

    def can_lift(self, value):
        return (self.num_levels == 1 and
                UF.bev_lift_supported(self.num_heads, self.embed_dims // self.num_heads,
                                      self.num_points, value.dtype))

    def project_value(self, value, key_padding_mask=None, passthru=False):
        if passthru:
            assert key_padding_mask is None
            value, alias = linear_pass(value, self.value_proj.weight, self.value_proj.bias)
            return _store_value(value), alias
        value = ubv_linear(value, self.value_proj.weight, self.value_proj.bias)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        return _store_value(value)

    def project_value_chained(self, value, chain, key_padding_mask=None):
        """``project_value`` of a tensor that EVERY layer of the encoder projects (the camera / LiDAR features).
        ``chain``: a dict the encoder hands to all its layers for one pass.  Layer k projects the pass-through alias
        layer k - 1 left there (``linear.linear_pass``: the same values), so the layers' input gradients arrive as a
        chain — each Linear's dX GEMM takes the later layers' sum as its accumulator input — instead of three
        tensors that autograd adds with two element-wise kernels over the feature map."""
        if chain is None or key_padding_mask is not None or not (value.is_cuda and torch.is_grad_enabled()
                                                                 and value.requires_grad):
            return self.project_value(value, key_padding_mask)
        prev = chain.get('alias')
        src = prev if prev is not None and prev.shape == value.shape and prev.dtype == value.dtype else value
        out, alias = self.project_value(src, passthru=True)
        chain['alias'] = alias
        return out

    def k1(self, value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
        bs, num_value = value.shape[:2]
        return UF.MultiScaleDeformableAttnFunction.apply(
            value.view(bs, num_value, self.num_heads, -1), spatial_shapes, level_start_index,
            sampling_locations, attention_weights, self.im2col_step)


@ATTENTION.register_module(name=['MultiScaleDeformableAttention', 'CustomMSDeformableAttention'])
class MultiScaleDeformableAttention(_DeformAttnBase):
    """Deformable self/cross attention with output projection, dropout and residual
    (decoder.py:230-338)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(embed_dims, num_heads, num_levels, num_points, im2col_step, batch_first,
                         norm_cfg, init_cfg, with_output_proj=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag='decoder', **kwargs):
        if 'residual' in kwargs and identity is None:         # deprecated_api_warning mapping
            identity = kwargs.pop('residual')
        if value is None:
            value = query
        if identity is None:
            identity = query
        pos_term = kwargs.pop('pos_term', None)
        if pos_term is not None and kwargs.get('return_parts') and self.batch_first and value is query and \
                identity is query and key_padding_mask is None and query.is_cuda and reference_points is not None and \
                reference_points.shape[-1] == 2 and reference_points.shape[2] == 1 and \
                pos_term.shape[0] == query.shape[1] and \
                UF.bev_lift_supported(self.num_heads, self.embed_dims // self.num_heads, self.num_points, query.dtype):
            # BEV self-attention with the positional term folded into the offset / logit GEMM (no `query + query_pos`):
            # query -> value_proj -> alias -> offsets | logits -> alias = the residual, so the residual's gradient
            # passes through both input-gradient GEMMs' epilogues and nothing is summed at the fan-out.  The first
            # layer's batch-expanded queries are materialised once, for both GEMMs and the residual.
            x = query if query.is_contiguous() else query.contiguous()
            bs, num_query, _ = x.shape
            hw = static_hw(spatial_shapes)
            assert sum(h * w for h, w in hw) == num_query
            if self.value_proj.bias is not None and self.sampling_offsets.bias is not None and \
                    self_attn_in_supported(x, pos_term, self.value_proj.weight, self.sampling_offsets.weight,
                                           self.attention_weights.weight):
                # value_proj | sampling_offsets | attention_weights in one GEMM (x read once), their input and
                # weight gradients in one GEMM each
                wo, wa = self._ol_weights()
                v, offlog, identity = self_attn_in(x, pos_term, self.value_proj.weight, self.value_proj.bias,
                                                   wo, self.sampling_offsets.bias, wa, self.attention_weights.bias)
                v = _store_value(v)
            else:
                v, alias = self.project_value(x, passthru=True)
                offlog, identity = self.offsets_and_logits(alias, passthru=True, row_bias=pos_term)
            grid = kwargs.get('bev_h'), kwargs.get('bev_w')
            qgrid = grid if (grid[0] and grid[1] and grid[0] * grid[1] == num_query) else None
            output = UF.bev_lift(v, offlog, reference_points.reshape(1, bs, num_query, 1, 2), 1, hw[0],
                                 self.num_heads, self.num_points, query_grid=qgrid,
                                 ref_is_grid=bool(kwargs.get('ref_is_grid')) and qgrid is not None,
                                 slot_center=self.sampling_offsets.bias)
            output = ubv_linear(output, self.output_proj.weight, self.output_proj.bias)
            return output, identity, self.dropout.p
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        hw = static_hw(spatial_shapes)
        assert sum(h * w for h, w in hw) == num_value
        if kwargs.get('return_parts') and self.batch_first and identity is value and \
                key_padding_mask is None and value.is_cuda:
            # the layer input feeds value_proj AND the residual: route the residual through the
            # Linear's pass-through so that its gradient joins the input-gradient GEMM
            value, identity = self.project_value(value, passthru=True)
        else:
            value = self.project_value(value, key_padding_mask)
        H, L, P = self.num_heads, self.num_levels, self.num_points
        if reference_points.shape[-1] == 2 and self.can_lift(value) and \
                reference_points.shape[2] == 1:
            grid = kwargs.get('bev_h'), kwargs.get('bev_w')
            qgrid = grid if (grid[0] and grid[1] and grid[0] * grid[1] == num_query) else None
            output = UF.bev_lift(value, self.offsets_and_logits(query),
                                 reference_points.reshape(1, bs, num_query, 1, 2), 1, hw[0], H, P,
                                 query_grid=qgrid,
                                 ref_is_grid=bool(kwargs.get('ref_is_grid')) and qgrid is not None,
                                 slot_center=self.sampling_offsets.bias)
        else:
            sampling_offsets = self.sampling_offsets(query).view(bs, num_query, H, L, P, 2)
            attention_weights = self.attention_weights(query).view(bs, num_query, H, L * P)
            attention_weights = attention_weights.softmax(-1).view(bs, num_query, H, L, P)
            if reference_points.shape[-1] == 2:
                offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
                sampling_locations = reference_points[:, :, None, :, None, :] \
                    + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
            elif reference_points.shape[-1] == 4:
                sampling_locations = reference_points[:, :, None, :, None, :2] \
                    + sampling_offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
            else:
                raise ValueError(f'Last dim of reference_points must be 2 or 4, but get '
                                 f'{reference_points.shape[-1]} instead.')
            output = self.k1(value, spatial_shapes, level_start_index, sampling_locations,
                             attention_weights)
        output = ubv_linear(output, self.output_proj.weight, self.output_proj.bias)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        if kwargs.get('return_parts'):            # the caller fuses dropout + residual + LayerNorm
            return output, identity, self.dropout.p
        return self.dropout(output) + identity


CustomMSDeformableAttention = MultiScaleDeformableAttention


class _MSDeformableAttention3D(_DeformAttnBase):
    """Z-anchored deformable sampling without output projection / residual
    (spatial_cross_attention_img.py:313-442 == spatial_cross_attention_pts.py:306-449)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__(embed_dims, num_heads, num_levels, num_points, im2col_step, batch_first,
                         norm_cfg, init_cfg, with_output_proj=False)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        hw = static_hw(spatial_shapes)
        assert sum(h * w for h, w in hw) == num_value
        value = self.project_value_chained(value, kwargs.get('value_chain'), key_padding_mask)
        H, L, P = self.num_heads, self.num_levels, self.num_points
        if reference_points.shape[-1] != 2:
            if reference_points.shape[-1] == 4:
                assert False
            raise ValueError(f'Last dim of reference_points must be 2 or 4, but get '
                             f'{reference_points.shape[-1]} instead.')
        num_Z_anchors = reference_points.shape[2]
        assert P % num_Z_anchors == 0
        alias = None
        want_alias = bool(kwargs.get('want_query_alias')) and query_pos is None and \
            self.batch_first and query.is_cuda
        if self.can_lift(value):
            offlog = self.offsets_and_logits(query, passthru=want_alias)
            if want_alias:
                offlog, alias = offlog
            output = UF.bev_lift(value, offlog,
                                 reference_points.reshape(1, bs, num_query, num_Z_anchors, 2), 1,
                                 hw[0], H, P, query_grid=kwargs.get('query_grid'),
                                 ref_is_grid=bool(kwargs.get('ref_is_grid')),
                                 slot_center=self.sampling_offsets.bias)
        else:
            sampling_offsets = self.sampling_offsets(query).view(bs, num_query, H, L, P, 2)
            attention_weights = self.attention_weights(query).view(bs, num_query, H, L * P)
            attention_weights = attention_weights.softmax(-1).view(bs, num_query, H, L, P)
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            sampling_offsets = sampling_offsets / offset_normalizer[None, None, None, :, None, :]
            sampling_offsets = sampling_offsets.view(bs, num_query, H, L, P // num_Z_anchors,
                                                     num_Z_anchors, 2)
            sampling_locations = reference_points[:, :, None, None, None, :, :] + sampling_offsets
            sampling_locations = sampling_locations.view(bs, num_query, H, L, P, 2)
            output = self.k1(value, spatial_shapes, level_start_index, sampling_locations,
                             attention_weights)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        if kwargs.get('want_query_alias'):        # (output, alias of query or None)
            return output, alias
        return output


@ATTENTION.register_module(name=['MSDeformableAttention3DImg',
                                 'MSDeformableAttention3DUniQueryImg'])
class MSDeformableAttention3DImg(_MSDeformableAttention3D):
    pass


@ATTENTION.register_module()
class MSDeformableAttention3DPts(_MSDeformableAttention3D):
    pass
