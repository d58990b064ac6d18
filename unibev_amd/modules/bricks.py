"""Transformer scaffolding with the constructor semantics the reference's configs rely on.

These restate the public behaviour of the [ext] mmcv-full 1.3.17 / mmdet 2.19.0 classes the
reference subclasses or builds from config: ``BaseModule``, ``FFN``,
``BaseTransformerLayer`` (constructor + the generic forward used by the decoder layer),
``TransformerLayerSequence``, ``LearnedPositionalEncoding``.  Attribute names (``attentions``,
``ffns``, ``norms``, ``layers``, ``row_embed`` ...) are state-dict keys of the published
checkpoints and must not change.
"""
import copy
import warnings

import torch
import torch.nn as nn

from .. import functional as UF
from ..linear import linear as ubv_linear
from ..linear import linear_pass, linear_relu_dropout, linear_after_relu_dropout
from ..registry import (ATTENTION, FEEDFORWARD_NETWORK, POSITIONAL_ENCODING, TRANSFORMER_LAYER,
                        TRANSFORMER_LAYER_SEQUENCE, build_attention, build_feedforward_network,
                        build_transformer_layer)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    """mmcv.cnn.xavier_init; a no-op on objects without weight/bias (``xavier_init(None)`` in the
    reference, spatial_cross_attention_img.py:310, is harmless)."""
    assert distribution in ('uniform', 'normal')
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


@FEEDFORWARD_NETWORK.register_module()
class FFN(BaseModule):
    """x + drop(W2 drop(act(W1 x))); state-dict keys ``layers.0.0.*`` and ``layers.1.*``."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2, f'num_fcs should be no less than 2. got {num_fcs}.'
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        act = (act_cfg or {}).get('type', 'ReLU')
        if act not in ('ReLU', 'GELU'):
            raise KeyError(f'{act} is not a supported FFN activation')
        layers, in_channels = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(
                nn.Linear(in_channels, feedforward_channels),
                nn.ReLU(inplace=True) if act == 'ReLU' else nn.GELU(),
                nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        p = (dropout_layer or {}).get('drop_prob', 0.) if dropout_layer else 0.
        self.dropout_layer = nn.Dropout(p) if p > 0 else nn.Identity()
        self.add_identity = add_identity

    def _mlp(self, x):
        """``self.layers(x)`` with the Linear layers routed through the split-K-wgrad linear."""
        for layer in self.layers:
            x = self._block(x, layer)
        return x

    def _block(self, x, layer, start=0):
        """One entry of ``self.layers``: a Linear, the closing Dropout, or a
        Sequential(Linear, act, Dropout) whose ReLU + Dropout run as one kernel on the GPU."""
        if isinstance(layer, nn.Linear):
            p_act = getattr(x, '_ubv_ffn_act', None)
            if p_act is not None:      # x = linear_relu_dropout(...): its derivative folds into this dgrad
                return linear_after_relu_dropout(x, layer.weight, layer.bias, p_act, True)
            return ubv_linear(x, layer.weight, layer.bias)
        if not isinstance(layer, nn.Sequential):
            return layer(x)
        subs = list(layer)
        i = start
        while i < len(subs):
            sub = subs[i]
            if isinstance(sub, nn.Linear) and self._is_relu_dropout(subs, i + 1) and x.is_cuda:
                # Linear -> ReLU -> Dropout: the activation runs in the GEMM epilogue
                x = linear_relu_dropout(x, sub.weight, sub.bias, subs[i + 2].p, self.training)
                x._ubv_ffn_act = subs[i + 2].p if self.training else 0.0
                i += 2
            elif isinstance(sub, nn.Linear):
                x = ubv_linear(x, sub.weight, sub.bias)
            elif isinstance(sub, nn.ReLU):
                nxt = subs[i + 1] if i + 1 < len(subs) else None
                if x.is_cuda and isinstance(nxt, nn.Dropout) and x.numel() % 8 == 0:
                    x = UF.relu_dropout(x, nxt.p, self.training)
                    i += 1
                else:
                    x = torch.relu(x)              # out-of-place: x is the output of a custom op
            else:
                x = sub(x)
            i += 1
        return x

    @staticmethod
    def _is_relu_dropout(subs, i):
        import os
        if os.environ.get('UBV_FFN_FUSE', '1') == '0':       # study knob: the separate relu_dropout kernels
            return False
        return i + 1 < len(subs) and isinstance(subs[i], nn.ReLU) and isinstance(subs[i + 1], nn.Dropout)

    def forward_parts(self, x, identity=None):
        """(pre-dropout output, identity, p) for a caller that fuses the tail with the next norm;
        None when this FFN's tail is not ``identity + Dropout(out)``."""
        last = self.layers[-1]
        if not (self.add_identity and isinstance(last, nn.Dropout) and
                isinstance(self.dropout_layer, nn.Identity)):
            return None
        h = x
        blocks = list(self.layers)[:-1]
        first = blocks[0] if blocks else None
        if identity is None and x.is_cuda and isinstance(first, nn.Sequential) and \
                isinstance(first[0], nn.Linear):
            # x feeds the first Linear AND the residual: pass-through (linear.linear_pass)
            if self._is_relu_dropout(list(first), 1):
                h, identity = linear_relu_dropout(x, first[0].weight, first[0].bias, first[2].p, self.training,
                                                  passthru=True)
                h._ubv_ffn_act = first[2].p if self.training else 0.0
                h = self._block(h, first, start=3)
            else:
                h, identity = linear_pass(x, first[0].weight, first[0].bias)
                h = self._block(h, first, start=1)
            blocks = blocks[1:]
        for layer in blocks:
            h = self._block(h, layer)
        return h, (x if identity is None else identity), last.p

    def forward(self, x, identity=None):
        out = self._mlp(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


@ATTENTION.register_module()
class MultiheadAttention(BaseModule):
    """mmcv wrapper of nn.MultiheadAttention (decoder self-attention slot):
    ``identity + dropout(attn(q + q_pos, k + k_pos, v))``; state-dict keys ``attn.*``."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=dict(type='Dropout', drop_prob=0.), init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__(init_cfg)
        if 'dropout' in kwargs:
            attn_drop = kwargs.pop('dropout')
            dropout_layer = dict(type='Dropout', drop_prob=attn_drop)
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        p = (dropout_layer or {}).get('drop_prob', 0.)
        self.dropout_layer = nn.Dropout(p) if p > 0 else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


@TRANSFORMER_LAYER.register_module()
class BaseTransformerLayer(BaseModule):
    """Builds ``attentions`` / ``ffns`` / ``norms`` from config exactly as mmcv does: deprecated
    kwargs ``feedforward_channels``/``ffn_dropout``/``ffn_num_fcs`` are folded into ``ffn_cfgs``;
    ``batch_first`` is injected into every attention config that lacks it."""

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, num_fcs=2,
                               ffn_drop=0., act_cfg=dict(type='ReLU', inplace=True)),
                 operation_order=None, norm_cfg=dict(type='LN'), init_cfg=None, batch_first=False,
                 **kwargs):
        deprecated_args = dict(feedforward_channels='feedforward_channels',
                               ffn_dropout='ffn_drop', ffn_num_fcs='num_fcs')
        ffn_cfgs = copy.deepcopy(ffn_cfgs)
        for ori_name, new_name in deprecated_args.items():
            if ori_name in kwargs:
                ffn_cfgs[new_name] = kwargs[ori_name]
        super().__init__(init_cfg)
        self.batch_first = batch_first
        assert set(operation_order) & {'self_attn', 'norm', 'ffn', 'cross_attn'} == \
            set(operation_order), \
            f'The operation_order of {self.__class__.__name__} should contains all four ' \
            f"operation type ['self_attn', 'norm', 'ffn', 'cross_attn']"
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs), \
                f'The length of attn_cfg {num_attn} is not consistent with the number of ' \
                f'attentionin operation_order {operation_order}.'
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = nn.ModuleList()
        index = 0
        for operation_name in operation_order:
            if operation_name in ('self_attn', 'cross_attn'):
                cfg = copy.deepcopy(dict(attn_cfgs[index]))
                if 'batch_first' in cfg:
                    assert self.batch_first == cfg['batch_first']
                else:
                    cfg['batch_first'] = self.batch_first
                attention = build_attention(cfg)
                attention.operation_name = operation_name
                self.attentions.append(attention)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        num_ffns = operation_order.count('ffn')
        if isinstance(ffn_cfgs, dict):
            ffn_cfgs = [copy.deepcopy(dict(ffn_cfgs)) for _ in range(num_ffns)]
        assert len(ffn_cfgs) == num_ffns
        for i in range(num_ffns):
            if 'embed_dims' not in ffn_cfgs[i]:
                ffn_cfgs[i]['embed_dims'] = self.embed_dims
            else:
                assert ffn_cfgs[i]['embed_dims'] == self.embed_dims
            self.ffns.append(build_feedforward_network(dict(ffn_cfgs[i]), dict(type='FFN')))
        norm_type = (norm_cfg or {}).get('type', 'LN')
        if norm_type != 'LN':
            raise KeyError(f'{norm_type} is not a supported norm layer (only LN)')
        self.norms = nn.ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(nn.LayerNorm(self.embed_dims))

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
            warnings.warn(f'Use same attn_mask in all attentions in {self.__class__.__name__} ')
        else:
            assert len(attn_masks) == self.num_attn
        for layer in self.operation_order:
            if layer == 'self_attn':
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value, identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[attn_index],
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


@TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """[ext] mmdet DetrTransformerDecoderLayer (decoder layer type named by the configs)."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == {'self_attn', 'norm', 'cross_attn', 'ffn'}


class TransformerLayerSequence(BaseModule):
    """``num_layers`` deep copies of one layer config as ``self.layers``."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        else:
            assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        for i in range(num_layers):
            self.layers.append(build_transformer_layer(dict(transformerlayers[i])))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


@POSITIONAL_ENCODING.register_module()
class LearnedPositionalEncoding(BaseModule):
    """[ext] mmdet LearnedPositionalEncoding as called at dense_heads/unibev_head.py:179-182:
    ``cat(col_embed(x), row_embed(y))`` -> (bs, 2*num_feats, h, w), x half first.

    Built token-major, so the (bs, C, h, w) result is a view whose later
    ``flatten(2).permute(2, 0, 1)`` (transformer_fusion.py:491-492) costs nothing."""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__(init_cfg)
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats = num_feats
        self.row_num_embed = row_num_embed
        self.col_num_embed = col_num_embed

    def init_weights(self):
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)
        self._is_init = True

    def forward(self, mask):
        bs, h, w = mask.shape[0], mask.shape[-2], mask.shape[-1]
        x_embed = self.col_embed.weight[:w]
        y_embed = self.row_embed.weight[:h]
        pos = torch.cat((x_embed.unsqueeze(0).expand(h, w, -1),
                         y_embed.unsqueeze(1).expand(h, w, -1)), dim=-1)      # (h, w, 2F)
        out = pos.permute(2, 0, 1).unsqueeze(0).expand(bs, -1, -1, -1)
        # the un-expanded (h, w, C) table: UniBEVTransformer folds `query + query_pos` of the BEV self-attentions into a
        # per-query bias of their offset / logit Linears (encoders._EncoderBase._fold_pos_terms), which needs the
        # positional term once, not once per sample
        out._ubv_pos_hw = pos
        return out


class _ExpandBatch(torch.autograd.Function):
    """``t.expand(bs, ...)`` of a (1, ...) tensor whose backward adds the bs slices with plain
    elementwise adds: the framework's outer-dimension reduction of a (2, 40000, 256) gradient took
    328 us against ~20 us for one add (profiles/r01_v9_*)."""

    @staticmethod
    def forward(ctx, t, bs):
        return t.expand(bs, *t.shape[1:])

    @staticmethod
    def backward(ctx, g):
        out = g[0] if g.shape[0] == 1 else g[0] + g[1]
        for b in range(2, g.shape[0]):
            out = out + g[b]
        return out.unsqueeze(0), None


def cast_keep_expand(t, dtype):
    """``t.to(dtype)`` that keeps a batch-expanded (stride-0 leading dim) tensor expanded: the cast
    touches one sample's worth of data and every later broadcast add reads it once, instead of
    materialising bs copies (and their transposed strides) as a plain ``.to`` would."""
    if t is None:
        return t
    base = getattr(t, '_ubv_pos_hw', None)                  # LearnedPositionalEncoding's un-expanded table rides along
    if t.dim() > 1 and t.stride(0) == 0 and t.shape[0] > 1:
        out = _ExpandBatch.apply(t[:1].to(dtype), t.shape[0])
    else:
        out = t.to(dtype)
    if base is not None and out is not t and base.dtype == dtype:
        out._ubv_pos_hw = base
    return out


__all__ = ['BaseModule', 'FFN', 'MultiheadAttention', 'BaseTransformerLayer',
           'DetrTransformerDecoderLayer', 'TransformerLayerSequence', 'LearnedPositionalEncoding',
           'xavier_init', 'constant_init', 'TRANSFORMER_LAYER_SEQUENCE', 'cast_keep_expand']
