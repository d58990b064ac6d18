"""Per-modality BEV encoders: ``ImgEncoder``/``ImgLayer`` and ``PtsEncoder``/``PtsLayer``.

Reference: models/modules/encoder_unibev_detr_img.py and encoder_unibev_detr_pts.py (same
registry keys, kwargs, forward signatures and state-dict layout).  Differences are mechanical:
the pillar reference grids are cached per shape instead of rebuilt every forward, the camera
projection + visibility runs as one HIP kernel (``functional.point_sampling``) whose per-camera
visibility / count by-products are handed to the cross-attention, and the (1,2) shape tensors of
the self-attention are built once per device.
"""
import copy
import warnings

import numpy as np
import torch

from .. import functional as UF
from ..registry import TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE
from .bricks import BaseTransformerLayer, TransformerLayerSequence
from .deform_attn import shapes_tensor

import os
_FOLD_POS = os.environ.get('UBV_FOLD_POS', '1') != '0'


def pillar_axes(H, W, Z, num_points_in_pillar, device, dtype=torch.float32):
    """Normalised 1-D pillar coordinates (xs[W], ys[H], zs[D]) with the reference's arithmetic
    (encoder_unibev_detr_img.py:68-73): ``linspace(0.5, n - 0.5, n) / n``."""
    # built on the host (torch CPU linspace is the parity target) and moved once; callers cache
    zs = torch.linspace(0.5, Z - 0.5, num_points_in_pillar, dtype=dtype) / Z
    xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype) / W
    ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype) / H
    xs, ys, zs = xs.to(device), ys.to(device), zs.to(device)
    return xs, ys, zs


def reference_points_3d(H, W, Z=8, num_points_in_pillar=4, bs=1, device='cuda',
                        dtype=torch.float):
    xs, ys, zs = pillar_axes(H, W, Z, num_points_in_pillar, device, dtype)
    D = num_points_in_pillar
    ref = torch.stack((xs.view(1, 1, W).expand(D, H, W), ys.view(1, H, 1).expand(D, H, W),
                       zs.view(D, 1, 1).expand(D, H, W)), -1)            # (D, H, W, 3)
    return ref.reshape(D, H * W, 3)[None].repeat(bs, 1, 1, 1)            # (bs, D, Nq, 3)


def reference_points_2d(H, W, bs=1, device='cuda', dtype=torch.float):
    xs, ys, _ = pillar_axes(H, W, 1, 1, device, dtype)
    ref = torch.stack((xs.view(1, W).expand(H, W), ys.view(H, 1).expand(H, W)), -1)
    return ref.reshape(1, H * W, 2).repeat(bs, 1, 1).unsqueeze(2)        # (bs, Nq, 1, 2)


class _EncoderBase(TransformerLayerSequence):
    def __init__(self, *args, pc_range=None, return_intermediate=False, dataset_type='nuscenes',
                 **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.pc_range = pc_range
        self.fp16_enabled = False
        self._ref_cache = {}

    @staticmethod
    def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim='3d', bs=1, device='cuda',
                             dtype=torch.float):
        """Reference points of SCA ('3d': (bs, D, Nq, 3)) and self-attention ('2d':
        (bs, Nq, 1, 2)); encoder_unibev_detr_img.py:45-109."""
        if dim == '3d':
            return reference_points_3d(H, W, Z, num_points_in_pillar, bs, device, dtype)
        if dim == '2d':
            return reference_points_2d(H, W, bs, device, dtype)
        return None

    def _cached(self, key, fn):
        v = self._ref_cache.get(key)
        if v is None:
            v = fn()
            if len(self._ref_cache) > 8:
                self._ref_cache.clear()
            self._ref_cache[key] = v
        return v

    def _fold_pos_terms(self, base, bev_query):
        """``bev_pos`` is the same for every sample, so ``(query + bev_pos) . W^T`` of each layer's
        sampling_offsets | attention_weights Linear is ``query . W^T + (bev_pos . W^T)[q]``: ONE GEMM over the (Nq, C)
        table for all layers of this encoder gives the second terms, which the layers' GEMMs add in their epilogue
        (``ubv_gemm_nt_rowbias``).  Per layer this removes the ``query + query_pos`` pass (and the saved sum), lets the
        residual's gradient ride through both input-gradient GEMMs of the self-attention (no gradient sum at the
        fan-out), and shrinks the positional gradient from (bs, Nq, C) accumulations to one (Nq, .) GEMM.
        Returns the per-layer (Nq, H*P*3) views, or None when a layer is not shaped for it.  UBV_FOLD_POS=0: off."""
        if base is None or not _FOLD_POS or not bev_query.is_cuda or bev_query.dtype != torch.float32:
            return None
        from .deform_attn import MultiScaleDeformableAttention
        ws, sizes = [], []
        for layer in self.layers:
            att = layer.attentions[0] if len(getattr(layer, 'attentions', ())) else None
            if not isinstance(att, MultiScaleDeformableAttention) or layer.operation_order[0] != 'self_attn' or \
                    layer.pre_norm or not att.batch_first or att.num_levels != 1 or \
                    att.sampling_offsets.weight.shape[1] != base.shape[1]:
                return None
            ws += [att.sampling_offsets.weight, att.attention_weights.weight]
            sizes.append(att.sampling_offsets.weight.shape[0] + att.attention_weights.weight.shape[0])
        # ONE GEMM over the table for all layers (``linear.pos_fold_all``): every layer's term is a column view of its output
        # and carries the slot its gradient goes to, so the backward is one input-gradient GEMM + one weight-gradient pass
        # over ONE gradient matrix — rounds 3 - 5 wrapped the same GEMMs in ``torch.split``, whose backward (a framework
        # cat, plus one framework add per weight with a second consumer) ran inside the two-stream window
        # (unibev_amd/debug.py).  The weights have two consumers — this fold and the layer's own GEMM: they part in
        # ``functional.fan_out_pair``, whose backward adds the two gradients with this library's kernel.
        # ``cut_after`` = k: the upper layers' terms come from their own fold over a severed leaf of the table
        # (graph_step.GraphedStep cuts the backward after layer k: the upper layers' weights must not sit in the lower
        # half's graph).  UBV_FOLD_CHAIN=0: round 5's form (A/B runs).
        k = int(getattr(self, 'cut_after', 0) or 0)
        if not _FOLD_CHAIN:
            from ..linear import linear_cat
            if 0 < k < len(self.layers):
                lo = linear_cat(base, ws[:2 * k], [None] * (2 * k))
                hi = linear_cat(self._sever(base), ws[2 * k:], [None] * (len(ws) - 2 * k))
                return list(torch.split(lo, sizes[:k], dim=1)) + list(torch.split(hi, sizes[k:], dim=1))
            return list(torch.split(linear_cat(base, ws, [None] * len(ws)), sizes, dim=1))
        from ..linear import pos_fold_all
        fan = torch.is_grad_enabled() and all(w.requires_grad for w in ws)
        fold_ws = []
        for li, layer in enumerate(self.layers):
            so, aw = ws[2 * li], ws[2 * li + 1]
            if fan:
                (so, so_layer), (aw, aw_layer) = UF.fan_out_pair(so, aw)
                layer.attentions[0]._ubv_w_alias = (so_layer, aw_layer)
            fold_ws += [so, aw]
        if 0 < k < len(self.layers):
            return pos_fold_all(base, fold_ws[:2 * k]) + pos_fold_all(self._sever(base), fold_ws[2 * k:])
        return pos_fold_all(base, fold_ws)

    def _sever(self, x):
        """``cut_after``: a tensor of the lower layers that the upper layers read is handed to them as a fresh LEAF (no
        copy); the (tensor, leaf) pair is kept in ``self._cuts`` so that the backward can be run in two parts — from the
        loss down to the leaves, then from the tensors on with the leaves' gradients (graph_step.GraphedStep)."""
        if not torch.is_tensor(x) or not x.requires_grad or not torch.is_grad_enabled():
            return x
        leaf = x.detach().requires_grad_()
        self._cuts.append((x, leaf))
        return leaf

    def _run_layers(self, bev_query, key, value, args, layer_kwargs):
        intermediate = []
        output = bev_query
        self._cuts = []
        cut = int(getattr(self, 'cut_after', 0) or 0)
        pos_terms = self._fold_pos_terms(layer_kwargs.pop('bev_pos_base', None), bev_query)
        # every layer's cross-attention projects the same features: their input gradients form a chain
        # (deform_attn.project_value_chained); a severed graph starts a new one
        layer_kwargs['value_chain'] = {} if _VALUE_CHAIN else None
        for li, layer in enumerate(self.layers):
            if cut and li == cut:
                layer_kwargs['value_chain'] = {} if _VALUE_CHAIN else None
                same = value is key
                bev_query = self._sever(bev_query)
                key = self._sever(key)
                value = key if same else self._sever(value)
                if layer_kwargs.get('bev_pos') is not None:
                    layer_kwargs['bev_pos'] = self._sever(layer_kwargs['bev_pos'])
            if pos_terms is not None:
                layer_kwargs['pos_term'] = pos_terms[li]
            if li == 1:
                layer_kwargs.pop('query_table', None)      # only the first layer's queries are the table itself
            output = layer(bev_query, key, value, *args, **layer_kwargs)
            bev_query = output
            if self.return_intermediate:
                intermediate.append(output)
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output


def _lidar2img_tensor(img_metas, device):
    """(B, Nc, 4, 4) float32 on ``device``.  Accepts the reference's numpy metas
    (encoder_unibev_detr_img.py:115-124: one H2D copy per forward) or device-resident tensors."""
    first = img_metas[0]['lidar2img']
    if torch.is_tensor(first):
        return torch.stack([m['lidar2img'] for m in img_metas]).to(device=device,
                                                                    dtype=torch.float32)
    if isinstance(first, (list, tuple)) and len(first) and torch.is_tensor(first[0]):
        return torch.stack([torch.stack(list(m['lidar2img'])) for m in img_metas]).to(
            device=device, dtype=torch.float32)
    arr = np.ascontiguousarray(np.asarray([m['lidar2img'] for m in img_metas]).astype(np.float32))
    # reference: reference_points.new_tensor(float64 array) -> rounds to f32, one upload per forward.
    # Content-addressed cache: the same matrices (evaluation loops, repeated samples, the bench) are
    # uploaded once; loaders that want no upload at all hand over device tensors
    # (unibev_amd.pipelines.metas_to_device).
    key = (str(device), arr.shape, arr.tobytes())
    hit = _L2I_CACHE.get(key)
    if hit is None:
        hit = torch.from_numpy(arr).to(device, non_blocking=True)
        if len(_L2I_CACHE) >= 64:
            _L2I_CACHE.pop(next(iter(_L2I_CACHE)))
        _L2I_CACHE[key] = hit
    return hit


_L2I_CACHE = {}


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class ImgEncoder(_EncoderBase):
    def __init__(self, *args, pc_range=None, num_points_in_pillar=4, return_intermediate=False,
                 dataset_type='nuscenes', **kwargs):
        super().__init__(*args, pc_range=pc_range, return_intermediate=return_intermediate,
                         dataset_type=dataset_type, **kwargs)
        self.num_points_in_pillar = num_points_in_pillar

    def point_sampling(self, reference_points, pc_range, img_metas, with_visibility=False):
        """encoder_unibev_detr_img.py:112-187 on the GPU.  ``reference_points`` is the (bs, D, Nq, 3)
        grid of ``get_reference_points``; returns reference_points_cam (Nc,B,Nq,D,2) and bev_mask
        (Nc,B,Nq,D) bool [+ (vis0, count) when ``with_visibility``]."""
        bs, D, Nq, _ = reference_points.shape
        rp = reference_points[0].float()
        # recover the separable axes of the grid: x varies fastest
        W = int((rp[0, :, 1] == rp[0, 0, 1]).sum().item()) if Nq > 1 else 1
        xs = rp[0, :W, 0].contiguous()
        ys = rp[0, ::W, 1].contiguous()
        zs = rp[:, 0, 2].contiguous()
        return self._project(xs, ys, zs, pc_range, img_metas, rp.device, with_visibility)

    def _project(self, xs, ys, zs, pc_range, img_metas, device, with_visibility, l2i=None):
        # (``l2i``: the batched matrices when the caller made them already — transformer._encode does, ahead of the
        #  two-stream fork: stacking per-sample device tensors is a framework kernel)
        if l2i is None:
            l2i = _lidar2img_tensor(img_metas, device)
        shape0 = img_metas[0]['img_shape'][0]
        cam, mask, vis0, count = UF.point_sampling(l2i, xs, ys, zs, pc_range,
                                                   (shape0[0], shape0[1]))
        if with_visibility:
            return cam, mask, vis0, count
        return cam, mask

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, **kwargs):
        """bev_query (Nq, bs, C); key/value (num_cam, sum hw, bs, C) -> (bs, Nq, C)."""
        bs, dev = bev_query.size(1), bev_query.device
        # reference points stay f32 even when the query stream is 16-bit: a bf16 (x+.5)/200
        # would move the sampling grid by up to a pixel
        dt = bev_query.dtype if bev_query.dtype in (torch.float32, torch.float64) else torch.float32
        Z = self.pc_range[5] - self.pc_range[2]
        D = self.num_points_in_pillar
        axes = self._cached(('axes', bev_h, bev_w, Z, D, dev),
                            lambda: pillar_axes(bev_h, bev_w, Z, D, dev))
        ref_3d = self._cached(('3d', bev_h, bev_w, Z, D, bs, dev, dt),
                              lambda: reference_points_3d(bev_h, bev_w, Z, D, bs, dev, dt))
        ref_2d = self._cached(('2d', bev_h, bev_w, bs, dev, dt),
                              lambda: reference_points_2d(bev_h, bev_w, bs, dev, dt))
        cam, mask, vis0, count = self._project(*axes, self.pc_range, kwargs['img_metas'], dev, True,
                                               l2i=kwargs.pop('lidar2img_tensor', None))
        lists = UF.compact_visible(vis0, bev_w)  # once per pass; every layer's backward walks them (tile by tile)
        table = getattr(bev_query, '_ubv_table', None)    # the un-expanded query table (transformer._encode)
        bev_query = bev_query.permute(1, 0, 2)
        if bev_pos is not None:
            bev_pos = bev_pos.permute(1, 0, 2)
        layer_kwargs = dict(kwargs, query_table=table, bev_pos=bev_pos, ref_2d=ref_2d, ref_3d=ref_3d, bev_h=bev_h,
                            bev_w=bev_w, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, reference_points_cam=cam,
                            bev_mask=mask, cam_vis0=vis0, cam_count=count, cam_lists=lists,
                            query_grid=(bev_h, bev_w), ref_is_grid=True)
        return self._run_layers(bev_query, key, value, args, layer_kwargs)


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class PtsEncoder(_EncoderBase):
    def __init__(self, *args, pc_range=None, num_points_in_pillar_lidar=1,
                 return_intermediate=False, dataset_type='nuscenes', **kwargs):
        super().__init__(*args, pc_range=pc_range, return_intermediate=return_intermediate,
                         dataset_type=dataset_type, **kwargs)
        self.num_points_in_pillar_lidar = num_points_in_pillar_lidar

    def point_sampling(self, reference_points):
        """encoder_unibev_detr_pts.py:105-127: (bs, D, Nq, 3) -> xy as (D, bs, Nq, 2) plus the
        in-range mask the caller discards (quirk q6)."""
        rp = reference_points.clone().permute(1, 0, 2, 3)
        lidar = rp[..., :2]
        mask = ((lidar[..., 1:2] > 0.0) & (lidar[..., 1:2] < 1.0)
                & (lidar[..., 0:1] < 1.0) & (lidar[..., 0:1] > 0.0))
        return lidar, mask.permute(1, 2, 0, 3).squeeze(-1)

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None,
                spatial_shapes=None, level_start_index=None, valid_ratios=None, prev_bev=None,
                shift=0., **kwargs):
        bs, dev = bev_query.size(1), bev_query.device
        # reference points stay f32 even when the query stream is 16-bit: a bf16 (x+.5)/200
        # would move the sampling grid by up to a pixel
        dt = bev_query.dtype if bev_query.dtype in (torch.float32, torch.float64) else torch.float32
        Z = self.pc_range[5] - self.pc_range[2]
        D = self.num_points_in_pillar_lidar
        ref_3d = self._cached(('3d', bev_h, bev_w, Z, D, bs, dev, dt),
                              lambda: reference_points_3d(bev_h, bev_w, Z, D, bs, dev, dt))
        ref_2d = self._cached(('2d', bev_h, bev_w, bs, dev, dt),
                              lambda: reference_points_2d(bev_h, bev_w, bs, dev, dt))
        lidar = self._cached(('lidar', bev_h, bev_w, Z, D, bs, dev, dt),
                             lambda: self.point_sampling(ref_3d)[0].contiguous())
        table = getattr(bev_query, '_ubv_table', None)    # the un-expanded query table (transformer._encode)
        bev_query = bev_query.permute(1, 0, 2)
        if bev_pos is not None:
            bev_pos = bev_pos.permute(1, 0, 2)
        layer_kwargs = dict(kwargs, query_table=table, bev_pos=bev_pos, ref_2d=ref_2d, ref_3d=ref_3d, bev_h=bev_h,
                            bev_w=bev_w, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, reference_points_lidar=lidar,
                            query_grid=(bev_h, bev_w), ref_is_grid=True)
        return self._run_layers(bev_query, key, value, args, layer_kwargs)


_FOLD_CHAIN = os.environ.get('UBV_FOLD_CHAIN', '1') != '0'        # 0: the positional fold as one GEMM for all layers (A/B runs)
_VALUE_CHAIN = os.environ.get('UBV_VALUE_CHAIN', '1') != '0'      # 0: the layers' feature-map gradients added by autograd (A/B runs)
_SHARE_FIRST = os.environ.get('UBV_SHARE_FIRST', '1') != '0'     # 0: the first self-attention per sample (A/B runs)


class _BevLayer(BaseTransformerLayer):
    """Post-norm layer ('self_attn','norm','cross_attn','norm','ffn','norm'); the loop follows
    encoder_unibev_detr_img.py:395-481 including its positional-encoding quirk q4: the
    self-attention receives ``bev_pos``, the cross-attention (attn index 1) receives
    ``query_pos`` = None."""

    cross_kw = ()

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'),
                 batch_first=True, ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, batch_first=batch_first,
                         **kwargs)
        self.fp16_enabled = False
        self._self_shapes = {}

    def _bev_shapes(self, bev_h, bev_w, device):
        key = (bev_h, bev_w, device)
        v = self._self_shapes.get(key)
        if v is None:
            v = (shapes_tensor([(bev_h, bev_w)], device),
                 torch.zeros(1, dtype=torch.long, device=device))
            self._self_shapes[key] = v
        return v

    def forward(self, query, key=None, value=None, bev_pos=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, ref_2d=None,
                ref_3d=None, bev_h=None, bev_w=None, mask=None, spatial_shapes=None,
                level_start_index=None, pos_term=None, query_table=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
            warnings.warn(f'Use same attn_mask in all attentions in {self.__class__.__name__} ')
        else:
            assert len(attn_masks) == self.num_attn, \
                f'The length of attn_masks {len(attn_masks)} must be equal to the number of ' \
                f'attention in operation_order {self.num_attn}'
        order = self.operation_order
        fuse_next = False                 # the previous op handed (out, identity, p) to this norm
        shared = 0                        # > 0: the pending (out, identity) are ONE sample's rows for `shared` samples
        for op_i, layer in enumerate(order):
            # post-norm layers: `dropout(out) + identity` of an attention / FFN and the LayerNorm
            # that follows run as ONE kernel (functional.add_dropout_layernorm)
            parts = (not self.pre_norm and op_i + 1 < len(order) and order[op_i + 1] == 'norm'
                     and query.is_cuda)
            if layer == 'self_attn':
                ss, lsi = self._bev_shapes(bev_h, bev_w, query.device)
                attn = self.attentions[attn_index]

                def run_self(q, pos, ref):
                    return attn(q, q, q, identity if self.pre_norm else None, query_pos=pos, key_pos=pos,
                                attn_mask=attn_masks[attn_index], key_padding_mask=query_key_padding_mask,
                                reference_points=ref, spatial_shapes=ss, level_start_index=lsi, bev_h=bev_h,
                                bev_w=bev_w, return_parts=parts, pos_term=pos_term, **kwargs)
                # The first layer's queries are ONE table for every sample (the reference repeats bev_query over the
                # batch, encoder_unibev_detr_img.py:228-232; here a stride-0 expand): value / offset / logit
                # projections, the sampling and output_proj of its self-attention are the same rows bs times.  They run
                # for one sample; the samples part at the dropout of the fused add + LayerNorm, which reads the shared
                # rows with a row period (ubv_add_dropout_layernorm_*: bcast_rows).  Same values, 1 / bs of the work.
                bs = query.shape[0] if query.dim() == 3 else 1
                one = (parts and _SHARE_FIRST and bs > 1 and query.stride(0) == 0 and attn_masks[attn_index] is None
                       and query_key_padding_mask is None and (ref_2d is None or ref_2d.shape[0] == bs)
                       and (pos_term is not None or bev_pos is None or bev_pos.shape[0] == 1
                            or bev_pos.stride(0) == 0))      # (the positional term must be one table too)
                out = None
                if one:
                    # the table itself when the caller handed it over (transformer._encode): a slice of the expanded
                    # queries would send its gradient through a zero-filled [bs, Nq, C] tensor and a sum over the batch
                    q1 = query[:1]
                    if query_table is not None and query_table.shape == query.shape[1:] and \
                            query_table.dtype == query.dtype:
                        q1 = query_table.unsqueeze(0)
                    out = run_self(q1, None if bev_pos is None else bev_pos[:1],   # (unused under pos_term)
                                   None if ref_2d is None else ref_2d[:1])
                    if not (isinstance(out, tuple) and out[0].shape[0] == 1 and out[1].shape[0] == 1):
                        out = None                             # (not the fused form: every sample on its own)
                shared = bs if out is not None else 0
                query = out if out is not None else run_self(query, bev_pos, ref_2d)
                attn_index += 1
                fuse_next = parts and isinstance(query, tuple)
                if not fuse_next:
                    identity = query
            elif layer == 'norm':
                norm = self.norms[norm_index]
                if fuse_next:
                    out, res, p = query
                    query = UF.add_dropout_layernorm(out, res, norm.weight, norm.bias, p,
                                                     self.training, norm.eps, batch=shared)
                    identity = query
                    fuse_next, shared = False, 0
                else:
                    query = norm(query)
                norm_index += 1
            elif layer == 'cross_attn':
                pos = (bev_pos, bev_pos) if attn_index == 0 else (query_pos, key_pos)
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=pos[0],
                    key_pos=pos[1], reference_points=ref_3d, mask=mask,
                    attn_mask=attn_masks[attn_index], key_padding_mask=key_padding_mask,
                    spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                    return_parts=parts, **kwargs)
                attn_index += 1
                fuse_next = parts and isinstance(query, tuple)
                if not fuse_next:
                    identity = query
            elif layer == 'ffn':
                fp = self.ffns[ffn_index].forward_parts(query) if parts else None
                if fp is not None:
                    query, fuse_next = fp, True
                else:
                    query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


@TRANSFORMER_LAYER.register_module()
class ImgLayer(_BevLayer):
    pass


@TRANSFORMER_LAYER.register_module()
class PtsLayer(_BevLayer):
    pass
