"""Container-only stand-ins for the un-vendored third-party symbols the reference imports.

Used ONLY by ``make_golden.py`` to import ``/root/reference`` modules on CPU and record
golden input/output vectors.  Nothing here ships in the product path and nothing here is
reference source: it restates the *public* behaviour of mmcv-full 1.3.17 / mmdet 2.19.0
(pinned by the reference in docs/installation.md:6-9) for the handful of symbols the
hot-path modules touch.
"""
import copy
import functools
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- registry
class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args or None)


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f'{typ} is not in the {registry.name} registry')
    return cls(**args)


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


ATTENTION = Registry('attention')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
FEEDFORWARD_NETWORK = Registry('feed-forward Network')
TRANSFORMER = Registry('Transformer')


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


# --------------------------------------------------------------------------- init helpers
def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


# --------------------------------------------------------------------------- runner
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _identity_decorator_factory(*dargs, **dkwargs):
    def deco(fn):
        return fn
    return deco


def deprecated_api_warning(name_dict, cls_name=None):
    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            for old, new in name_dict.items():
                if old in kwargs:
                    kwargs[new] = kwargs.pop(old)
            return fn(*args, **kwargs)
        return wrapped
    return deco


def digit_version(v):
    out = []
    for p in v.split('+')[0].split('.'):
        num = ''.join(ch for ch in p if ch.isdigit())
        out.append(int(num) if num else 0)
    return tuple(out)


def to_2tuple(x):
    return (x, x) if not isinstance(x, (tuple, list)) else tuple(x)


class _ExtLoader:
    @staticmethod
    def load_ext(name, funcs):
        return types.SimpleNamespace(**{f: None for f in funcs})


# --------------------------------------------------------------------------- MSDA op
def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations,
                                        attention_weights):
    """Public mmcv CPU definition: per level grid_sample(bilinear, zeros,
    align_corners=False) on 2*loc-1, weighted sum over levels*points."""
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([int(h) * int(w) for h, w in value_spatial_shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(value_spatial_shapes):
        h, w = int(h), int(w)
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * num_heads, embed_dims, h, w)
        g = sampling_grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))
    attention_weights = attention_weights.transpose(1, 2).reshape(
        bs * num_heads, 1, num_queries, num_levels * num_points)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * attention_weights).sum(-1)
    return out.view(bs, num_heads * embed_dims, num_queries).transpose(1, 2).contiguous()


class MultiScaleDeformableAttnFunction:
    @staticmethod
    def apply(value, spatial_shapes, level_start_index, sampling_locations,
              attention_weights, im2col_step):
        return multi_scale_deformable_attn_pytorch(value, spatial_shapes, sampling_locations,
                                                   attention_weights)


# --------------------------------------------------------------------------- transformer bricks
class FFN(BaseModule):
    """mmcv FFN: x + drop(W2 drop(relu(W1 x)))."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        layers = []
        in_ch = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(in_ch, feedforward_channels),
                                     nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
            in_ch = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


FEEDFORWARD_NETWORK.register_module()(FFN)


class BaseTransformerLayer(BaseModule):
    """Constructor semantics of mmcv 1.3.17 BaseTransformerLayer (forward is overridden by
    the reference's ImgLayer / PtsLayer)."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024,
                               num_fcs=2, ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None,
                 batch_first=False, **kwargs):
        deprecated = dict(feedforward_channels='feedforward_channels',
                          ffn_dropout='ffn_drop', ffn_num_fcs='num_fcs')
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for ori, new in deprecated.items():
            if ori in kwargs:
                ffn_cfgs[new] = kwargs[ori]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        index = 0
        for name in operation_order:
            if name in ('self_attn', 'cross_attn'):
                cfg = copy.deepcopy(attn_cfgs[index])
                if 'batch_first' in cfg:
                    assert self.batch_first == cfg['batch_first']
                else:
                    cfg['batch_first'] = self.batch_first
                attention = build_attention(cfg)
                attention.operation_name = name
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(num_ffns)]
        for i in range(num_ffns):
            if 'embed_dims' not in ffn_cfgs[i]:
                ffn_cfgs[i]['embed_dims'] = self.embed_dims
            self.ffns.append(build_feedforward_network(ffn_cfgs[i]))
        self.norms = ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(nn.LayerNorm(self.embed_dims))


def _base_layer_forward(self, query, key=None, value=None, query_pos=None, key_pos=None,
                        attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, **kwargs):
    """mmcv 1.3.17 BaseTransformerLayer.forward: walk ``operation_order``; every attention / FFN
    adds its own residual (``identity`` is only handed over in pre-norm layers)."""
    norm_i = attn_i = ffn_i = 0
    identity = query
    masks = [None] * self.num_attn if attn_masks is None else (
        [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        if isinstance(attn_masks, torch.Tensor) else attn_masks)
    for op in self.operation_order:
        if op == 'self_attn':
            query = self.attentions[attn_i](
                query, query, query, identity if self.pre_norm else None, query_pos=query_pos,
                key_pos=query_pos, attn_mask=masks[attn_i], key_padding_mask=query_key_padding_mask,
                **kwargs)
            attn_i += 1
            identity = query
        elif op == 'cross_attn':
            query = self.attentions[attn_i](
                query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                key_pos=key_pos, attn_mask=masks[attn_i], key_padding_mask=key_padding_mask, **kwargs)
            attn_i += 1
            identity = query
        elif op == 'norm':
            query = self.norms[norm_i](query)
            norm_i += 1
        elif op == 'ffn':
            query = self.ffns[ffn_i](query, identity if self.pre_norm else None)
            ffn_i += 1
    return query


BaseTransformerLayer.forward = _base_layer_forward


class MultiheadAttention(BaseModule):
    """mmcv 1.3.17 ``MultiheadAttention``: nn.MultiheadAttention with positional encodings added to
    query / key, optional batch_first, and ``identity + dropout(proj_drop(out))``.  The deprecated
    ``dropout`` kwarg sets both the attention dropout and the output dropout."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__(init_cfg)
        dropout_layer = dict(dropout_layer) if dropout_layer else None
        if 'dropout' in kwargs:
            attn_drop = kwargs.pop('dropout')
            dropout_layer = dict(type='Dropout', drop_prob=attn_drop)
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(dropout_layer['drop_prob']) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        q = query if query_pos is None else query + query_pos
        k = key if key_pos is None else key + key_pos
        if self.batch_first:
            q, k, value = q.transpose(0, 1), k.transpose(0, 1), value.transpose(0, 1)
        out = self.attn(query=q, key=k, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


ATTENTION.register_module()(MultiheadAttention)


class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """mmdet 2.19.0 ``DetrTransformerDecoderLayer``: a BaseTransformerLayer whose six operations
    must be {self_attn, norm, cross_attn, ffn}."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == set(['self_attn', 'norm', 'cross_attn', 'ffn'])


TRANSFORMER_LAYER.register_module()(DetrTransformerDecoderLayer)


# --------------------------------------------------------------------------- head-side [ext] pieces
HEADS = Registry('head')
POSITIONAL_ENCODING = Registry('position encoding')


def inverse_sigmoid(x, eps=1e-5):
    """mmdet.models.utils.transformer.inverse_sigmoid."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def bias_init_with_prob(prior_prob):
    import math
    return float(-math.log((1 - prior_prob) / prior_prob))


class LearnedPositionalEncoding(BaseModule):
    """mmdet 2.19.0 ``LearnedPositionalEncoding``: x (column) half first."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__(init_cfg)
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x_embed = self.col_embed(torch.arange(w, device=mask.device))
        y_embed = self.row_embed(torch.arange(h, device=mask.device))
        pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)),
                        dim=-1).permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)
        return pos


POSITIONAL_ENCODING.register_module()(LearnedPositionalEncoding)


class DETRHead(BaseModule):
    """What mmdet 2.19.0 ``DETRHead.__init__`` leaves on ``self`` for a subclass that overrides
    ``_init_layers`` / ``forward`` (losses, assigner and sampler are not built: the head's
    forward never touches them)."""

    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None,
                 sync_cls_avg_factor=False, positional_encoding=None, loss_cls=None, loss_bbox=None,
                 loss_iou=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.num_query, self.num_classes, self.in_channels = num_query, num_classes, in_channels
        self.num_reg_fcs = num_reg_fcs
        self.sync_cls_avg_factor = sync_cls_avg_factor
        use_sigmoid = (loss_cls or {}).get('use_sigmoid', False)
        self.loss_cls = types.SimpleNamespace(use_sigmoid=use_sigmoid)
        self.cls_out_channels = num_classes if use_sigmoid else num_classes + 1
        self.act_cfg = (transformer or {}).get('act_cfg', dict(type='ReLU', inplace=True))
        self.activate = nn.ReLU(inplace=True)
        self.positional_encoding = build_from_cfg(positional_encoding, POSITIONAL_ENCODING)
        self.transformer = build_from_cfg(transformer, TRANSFORMER)
        self.embed_dims = self.transformer.embed_dims
        self._init_layers()


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


class NullDecoder(TransformerLayerSequence):
    """Consumer placeholder: the decoder is out of scope."""

    def __init__(self, *args, **kwargs):
        BaseModule.__init__(self)
        self.num_layers = 0

    def forward(self, query, *args, reference_points=None, **kwargs):
        return query[None], reference_points[None]


TRANSFORMER_LAYER_SEQUENCE.register_module(name='NullDecoder')(NullDecoder)


# --------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Insert the stand-in modules into ``sys.modules`` (idempotent)."""
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_unibev_stub', False):
        return
    mmcv = _mod('mmcv', _unibev_stub=True, ConfigDict=ConfigDict)
    mmcv.__path__ = []
    cnn = _mod('mmcv.cnn', xavier_init=xavier_init, constant_init=constant_init)
    cnn.__path__ = []
    bricks = _mod('mmcv.cnn.bricks')
    bricks.__path__ = []
    _mod('mmcv.cnn.bricks.registry', ATTENTION=ATTENTION, TRANSFORMER_LAYER=TRANSFORMER_LAYER,
         TRANSFORMER_LAYER_SEQUENCE=TRANSFORMER_LAYER_SEQUENCE,
         FEEDFORWARD_NETWORK=FEEDFORWARD_NETWORK)
    _mod('mmcv.cnn.bricks.transformer', build_attention=build_attention,
         build_transformer_layer_sequence=build_transformer_layer_sequence,
         build_transformer_layer=build_transformer_layer,
         build_feedforward_network=build_feedforward_network,
         BaseTransformerLayer=BaseTransformerLayer,
         TransformerLayerSequence=TransformerLayerSequence, FFN=FFN)
    runner = _mod('mmcv.runner', force_fp32=_identity_decorator_factory,
                  auto_fp16=_identity_decorator_factory, BaseModule=BaseModule)
    runner.__path__ = []
    _mod('mmcv.runner.base_module', BaseModule=BaseModule, ModuleList=ModuleList,
         Sequential=Sequential)
    utils = _mod('mmcv.utils', TORCH_VERSION=torch.__version__, digit_version=digit_version,
                 ext_loader=_ExtLoader, ConfigDict=ConfigDict, build_from_cfg=build_from_cfg,
                 deprecated_api_warning=deprecated_api_warning, to_2tuple=to_2tuple,
                 Registry=Registry)
    utils.__path__ = []
    utils.path = _mod('mmcv.utils.path', mkdir_or_exist=lambda *a, **k: None)
    ops = _mod('mmcv.ops')
    ops.__path__ = []
    mmcv.cnn, mmcv.runner, mmcv.utils, mmcv.ops = cnn, runner, utils, ops
    cnn.bricks = bricks

    def _late_msda():
        # the self-attn slot: mmcv MultiScaleDeformableAttention == the copy the reference
        # vendors as CustomMSDeformableAttention; bound after import.
        return ATTENTION.get('CustomMSDeformableAttention')

    class _LazyMSDA:
        def __new__(cls, *a, **k):
            return _late_msda()(*a, **k)

    _mod('mmcv.ops.multi_scale_deform_attn',
         multi_scale_deformable_attn_pytorch=multi_scale_deformable_attn_pytorch,
         MultiScaleDeformableAttnFunction=MultiScaleDeformableAttnFunction,
         MultiScaleDeformableAttention=_LazyMSDA)
    for n in ('mmdet', 'mmdet.models', 'mmdet.models.utils'):
        _mod(n).__path__ = []
    _mod('mmdet.models.utils.builder', TRANSFORMER=TRANSFORMER)
    # head-side imports of models/dense_heads/unibev_head.py:6-22
    cnn.Linear = nn.Linear
    cnn.bias_init_with_prob = bias_init_with_prob
    mmcv.cnn = cnn
    _mod('mmdet.core', multi_apply=lambda f, *a, **k: tuple(map(list, zip(*map(f, *a)))),
         reduce_mean=lambda t: t)
    _mod('mmdet.models.utils.transformer', inverse_sigmoid=inverse_sigmoid)
    sys.modules['mmdet.models'].HEADS = HEADS
    _mod('mmdet.models.dense_heads', DETRHead=DETRHead)
    for n in ('mmdet3d', 'mmdet3d.core', 'mmdet3d.core.bbox', 'mmdet3d.unibev_plugin',
              'mmdet3d.unibev_plugin.core', 'mmdet3d.unibev_plugin.core.bbox'):
        _mod(n).__path__ = []
    _mod('mmdet3d.core.bbox.coders',
         build_bbox_coder=lambda cfg: types.SimpleNamespace(pc_range=cfg['pc_range']))
    _mod('mmdet3d.unibev_plugin.core.bbox.util', normalize_bbox=lambda *a, **k: None)
    _mod('numpy_unused')
    _mod('cv2')
    tv = _mod('torchvision')
    tv.__path__ = []
    tvt = _mod('torchvision.transforms')
    tvt.__path__ = []
    _mod('torchvision.transforms.functional', rotate=lambda *a, **k: None)
