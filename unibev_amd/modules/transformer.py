"""``UniBEVTransformer``: both BEV encoders, CNW / spatial normalisation and linear | avg | cat fusion.

Reference: models/modules/transformer_fusion.py:49-586 (registry key, constructor kwargs,
parameter names, forward signature and return tuple are the same).  The encoder+fusion part
(:463-549) is the hot path; the object-query decoder (:572-582) is a consumer that is built
when its config type is registered and skipped otherwise.

MI355X form of the glue:
  * ``_pre_process_*``: one LDS-tiled transpose kernel adds the camera / level embeddings while
    it flattens (``functional.flatten_embed``) and writes the [bs][cam][hw][C] layout the value
    projection reads, exposed as the reference's (num_cam, hw, bs, C) view.
  * channel/spatial normalisation + fusion + the final (bs,Nq,C)->(Nq,bs,C*s) permute: one kernel
    (``functional.bev_fuse``); the 2xC CNW softmax stays a tiny torch op so autograd reaches the
    ``*_channel_weights`` parameters.
  * modality dropout draws from ``np.random`` exactly like the reference (:227-228, 463-477), so a
    seeded run drops the same modalities.
"""
import os
import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import constant_, normal_

from .. import functional as UF
from ..linear import linear, lowp_step_cache
from ..registry import (TRANSFORMER, TRANSFORMER_LAYER_SEQUENCE,
                        build_transformer_layer_sequence)
from .bricks import BaseModule, cast_keep_expand, xavier_init
from .deform_attn import _DeformAttnBase, index_tensor, shapes_tensor

# learned per-(sample, channel) weights: Linear over the token axis of both modalities + this activation
# (transformer_fusion.py:136-151)
_MLP_NORMS = {'MLP_ChannelNormWeights': lambda: nn.ReLU(inplace=True),
              'Leaky_ReLU_MLP_ChannelNormWeights': lambda: nn.LeakyReLU(inplace=True),
              'ELU_MLP_ChannelNormWeights': lambda: nn.ELU(inplace=True),
              'Sigmoid_MLP_ChannelNormWeights': nn.Sigmoid}


class ModalityProjectionModule(BaseModule):
    """x + LayerNorm(relu(Linear(x))): the stand-in features of a missing modality, projected from the other one
    (transformer_fusion.py:26-47; ``net.0`` / ``net.2`` are checkpoint keys).  The Linear is the MFMA GEMM of the
    encoder layers; ReLU / LayerNorm / residual are framework element-wise ops — no shipped config builds this."""

    def __init__(self, embed_dims, with_norm=True, with_residual=True):
        super().__init__()
        layers = [nn.Linear(embed_dims, embed_dims), nn.ReLU(inplace=True)]
        if with_norm:
            layers.append(nn.LayerNorm(embed_dims))
        self.net = nn.Sequential(*layers)
        self.with_residual = with_residual

    def forward(self, x):
        out = torch.relu(linear(x, self.net[0].weight, self.net[0].bias))
        if len(self.net) > 2:
            out = self.net[2](out)
        return x + out if self.with_residual else out


import os as _os

# Image and point-cloud encoders on two HIP streams (UBV_TWO_STREAMS=0 or set_two_streams(False): one).
# Measured at bs = 2, L+C CNW: 95.8 -> 106.3 samples/s in fp32 (round 2), 131.0 -> 143.5 (round 4).
# Round 4 found the mode NOT reproducible as built in rounds 2 - 4 — gradients of the side branch 1e-2 off in part of the
# eager steps, the forward 1e-3 off in most HIP-graph replays, against 6e-6 for one stream — and traced it to the
# instruction level (profiles/r04_two_stream_race.txt): PACKED f32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 /
# v_pk_mul_f32, which clang's SLP vectoriser forms from pairs of scalar f32 operations) return wrong results in a wave
# while MFMA instructions of another kernel's wave run on the same SIMD — here: a lifting kernel's coordinate arithmetic
# beside the other encoder's GEMM.  The library is therefore built with -fno-slp-vectorize (csrc/Makefile: no packed
# f32 arithmetic in any kernel; not slower), after which the two-stream step reproduces like the one-stream step
# (tools/ab/grad_repro*.py, fwd_repro_graph.py: 0 of 60 replays / 30 eager runs differ;
# tests/test_bench_gpu.py::test_training_step_gradients_are_reproducible_eagerly_and_replayed runs both modes).
_QUERY_TABLE = os.environ.get('UBV_QUERY_TABLE', '1') != '0'
_TWO_STREAMS = [_os.environ.get('UBV_TWO_STREAMS', '1') != '0']
_REGION_HOOK = [None]       # debug mode (unibev_amd.debug.ForeignKernelLog): called with 'fork' / 'join' / 'bwd_fork' / 'bwd_mark'
_SIDE_STREAMS = {}


def set_two_streams(on):
    _TWO_STREAMS[0] = bool(on)


def _side_stream(device):
    key = device.index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[key]

@TRANSFORMER.register_module()
class UniBEVTransformer(BaseModule):
    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300,
                 img_encoder=None, pts_encoder=None, decoder=None, embed_dims=256,
                 use_cams_embeds=True, fusion_method='linear', drop_modality=None,
                 feature_norm=None, spatial_norm=None, use_modal_embeds=None, bev_h=200,
                 bev_w=200, dual_queries=False, vis_output=None, cna_constant_init=None,
                 **kwargs):
        super().__init__(**kwargs)
        if img_encoder is not None:
            self.img_bev_encoder = build_transformer_layer_sequence(img_encoder)
        if pts_encoder is not None:
            self.pts_bev_encoder = build_transformer_layer_sequence(pts_encoder)
        # The decoder consumes fused_bev_embed; it is outside the hot path.
        self.decoder = None
        if decoder is not None and decoder.get('type') in TRANSFORMER_LAYER_SEQUENCE:
            self.decoder = build_transformer_layer_sequence(decoder)
        self.dual_queries = dual_queries
        # residual stream of the encoders under autocast: 16-bit (the reference's fp16 convention) or f32
        self.lowp_stream = os.environ.get('UBV_STREAM', 'lowp') != 'fp32'
        self.embed_dims = embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.fp16_enabled = False
        self.cna_constant_norm = cna_constant_init
        self.bev_h = bev_h
        self.bev_w = bev_w
        self.use_cams_embeds = use_cams_embeds
        self.fusion_method = fusion_method
        if fusion_method in ('linear', 'avg'):
            self.scale_factor = 1
        elif fusion_method == 'cat':
            self.scale_factor = 2
        else:
            raise ValueError('Unrecognizable fusion method:{}'.format(fusion_method))
        self.drop_modality = drop_modality
        self.feature_norm = feature_norm
        self.spatial_norm = spatial_norm
        self.use_modal_embeds = use_modal_embeds
        self.two_stage_num_proposals = two_stage_num_proposals
        self.vis_output = vis_output
        self.l_flag = 1
        self.c_flag = 1
        self.init_layers()

    @property
    def with_img_bev_encoder(self):
        return getattr(self, 'img_bev_encoder', None) is not None

    @property
    def with_pts_bev_encoder(self):
        return getattr(self, 'pts_bev_encoder', None) is not None

    def init_layers(self):
        """Parameter names follow transformer_fusion.py:130-182 (they are checkpoint keys)."""
        if self.feature_norm == 'ChannelNormWeights':
            self.feature_norm_layer = nn.Softmax(dim=0)
            # (the reference allocates these with torch.Tensor(n) — uninitialised memory until init_weights(); zeros here:
            #  no RNG consumed, and a model used before init_weights() / load_state_dict() is finite instead of garbage)
            self.pts_channel_weights = nn.Parameter(torch.zeros(self.embed_dims))
            self.img_channel_weights = nn.Parameter(torch.zeros(self.embed_dims))
        elif self.feature_norm in _MLP_NORMS:
            self.channel_weights_proj = nn.Sequential(nn.Linear(self.bev_h * self.bev_w * 2, 2),
                                                      _MLP_NORMS[self.feature_norm]())
        elif self.feature_norm == 'ModalityProjection':
            assert self.fusion_method == 'cat'
            self.c_modal_proj = ModalityProjectionModule(self.embed_dims)
            self.l_modal_proj = ModalityProjectionModule(self.embed_dims)
        if self.spatial_norm == 'SpatialNormWeights':
            self.spatial_norm_layer = nn.Softmax(dim=0)
            self.pts_spatial_weights = nn.Parameter(torch.zeros(self.bev_h * self.bev_w))
            self.img_spatial_weights = nn.Parameter(torch.zeros(self.bev_h * self.bev_w))
        if self.with_img_bev_encoder:
            self.img_level_embeds = nn.Parameter(torch.zeros(self.num_feature_levels,
                                                              self.embed_dims))
            self.cams_embeds = nn.Parameter(torch.zeros(self.num_cams, self.embed_dims))
        if self.with_pts_bev_encoder:
            self.pts_level_embeds = nn.Parameter(torch.zeros(self.num_feature_levels,
                                                              self.embed_dims))
        if self.use_modal_embeds == 'MLP':
            self.modal_embbeding_mlp = nn.Sequential(nn.Linear(2, self.embed_dims // 2), nn.ReLU(inplace=True),
                                                     nn.Linear(self.embed_dims // 2, self.embed_dims),
                                                     nn.ReLU(inplace=True))
        elif self.use_modal_embeds == 'Fixed':
            self.modal_embbeding_C = nn.Parameter(torch.zeros(self.embed_dims))
            self.modal_embbeding_L = nn.Parameter(torch.zeros(self.embed_dims))
        self.reference_points = nn.Linear(self.embed_dims * self.scale_factor, 3)

    def init_weights(self):
        """transformer_fusion.py:184-225."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, _DeformAttnBase):
                m.init_weights()
        if self.with_pts_bev_encoder:
            normal_(self.pts_level_embeds)
        if self.with_img_bev_encoder:
            normal_(self.img_level_embeds)
            normal_(self.cams_embeds)
        if self.feature_norm == 'ChannelNormWeights':
            if self.cna_constant_norm is True:
                constant_(self.pts_channel_weights, 0.5)
                constant_(self.img_channel_weights, 0.5)
            else:
                normal_(self.pts_channel_weights)
                normal_(self.img_channel_weights)
        # (the reference also calls mmcv's xavier_init on the Sequential containers of the variants below — a no-op,
        #  a container has no ``weight`` — so their Linears keep the xavier_uniform_ of the first loop above)
        if self.spatial_norm == 'SpatialNormWeights':
            normal_(self.pts_spatial_weights)
            normal_(self.img_spatial_weights)
        xavier_init(self.reference_points, distribution='uniform', bias=0.)
        if self.use_modal_embeds == 'Fixed':
            normal_(self.modal_embbeding_C)
            normal_(self.modal_embbeding_L)

    def get_probability(self, prob):
        return True if np.random.random() < prob else False

    # -- feature flattening ---------------------------------------------------------------------
    def _pre_process_img_feats(self, mlvl_img_feats, bev_queries):
        """list[(bs, Nc, C, h, w)] -> ((Nc, sum hw, bs, C) view, spatial_shapes, level_start_index);
        transformer_fusion.py:230-255."""
        flat, shapes = [], []
        for lvl, feat in enumerate(mlvl_img_feats):
            bs, num_cam, c, h, w = feat.shape
            tok = UF.flatten_embed(feat.reshape(bs * num_cam, c, h * w),
                                   self.cams_embeds if self.use_cams_embeds else None,
                                   self.img_level_embeds, row=lvl)
            flat.append(tok.view(bs, num_cam, h * w, c))
            shapes.append((h, w))
        flat = flat[0] if len(flat) == 1 else torch.cat(flat, 2)
        spatial_shapes = shapes_tensor(shapes, bev_queries.device)
        starts = np.concatenate(([0], np.cumsum([h * w for h, w in shapes])[:-1]))
        level_start_index = index_tensor(starts, bev_queries.device)
        return flat.permute(1, 2, 0, 3), spatial_shapes, level_start_index

    def _pre_process_pts_feats(self, mlvl_pts_feats, bev_queries):
        """list[(bs, C, h, w)] -> ((sum hw, bs, C) view, ...); transformer_fusion.py:257-278.
        The reference concatenates levels on the channel axis (:272), which only works for one
        level; more than one level is rejected here instead of silently mis-shaped."""
        if len(mlvl_pts_feats) != 1:
            raise ValueError('pts features: the reference path supports exactly one level')
        feat = mlvl_pts_feats[0]
        bs, c, h, w = feat.shape
        tok = UF.flatten_embed(feat.reshape(bs, c, h * w), None, self.pts_level_embeds, row=0)
        spatial_shapes = shapes_tensor([(h, w)], bev_queries.device)
        level_start_index = index_tensor([0], bev_queries.device)
        return tok.permute(1, 0, 2), spatial_shapes, level_start_index

    # -- fusion ----------------------------------------------------------------------------------
    def _channel_factors(self, device):
        """Per-channel factors of (img, pts): CNW softmax (transformer_fusion.py:323-337) times the
        fusion rule's modality scalar (:282-302)."""
        C = self.embed_dims
        c, l = float(self.c_flag), float(self.l_flag)
        if self.fusion_method == 'avg':
            c, l = c / (self.c_flag + self.l_flag), l / (self.c_flag + self.l_flag)
        if self.feature_norm == 'ChannelNormWeights':
            fw = torch.stack((self.img_channel_weights, self.pts_channel_weights), 0)
            if self.c_flag == 1 and self.l_flag == 1:
                n = self.feature_norm_layer(fw)
                iw, pw = n[0], n[1]
            else:
                iw = self.feature_norm_layer(fw[0:1])[0]
                pw = self.feature_norm_layer(fw[1:2])[0]
            return iw * c, pw * l
        one = torch.ones(C, dtype=torch.float32, device=device)
        return one * c, one * l

    def _spatial_factors(self):
        if self.spatial_norm != 'SpatialNormWeights':
            return None, None
        sw = torch.stack((self.img_spatial_weights, self.pts_spatial_weights), 0)
        if self.c_flag == 1 and self.l_flag == 1:
            n = self.spatial_norm_layer(sw)
            return n[0], n[1]
        return self.spatial_norm_layer(sw[:1])[0], self.spatial_norm_layer(sw[1:])[0]

    def _learned_channel_weights(self, img, pts):
        """(bs, C) weights of (img, pts) from ``channel_weights_proj`` (transformer_fusion.py:345-358): a 2-way score
        per (sample, channel) from the 2 Nq tokens of that channel — two (2, Nq) x (Nq, C) products per sample."""
        lin = self.channel_weights_proj[0]
        nq = lin.in_features // 2
        w = lin.weight.to(img.dtype)
        score = torch.matmul(w[:, :nq], img) + torch.matmul(w[:, nq:], pts) + lin.bias.to(img.dtype)[:, None]
        score = self.channel_weights_proj[1](score.transpose(1, 2))           # (bs, C, 2)
        if self.c_flag == 1 and self.l_flag == 1:
            n = score.softmax(-1)
            return n[..., 0], n[..., 1]
        return score[..., :1].softmax(-1)[..., 0], score[..., 1:].softmax(-1)[..., 0]

    def _modal_embedding(self, ref):
        if self.use_modal_embeds == 'MLP':
            status = torch.tensor([float(self.c_flag), float(self.l_flag)], dtype=torch.float32, device=ref.device)
            return self.modal_embbeding_mlp(status)
        if self.use_modal_embeds == 'Fixed':
            return self.c_flag * self.modal_embbeding_C + self.l_flag * self.modal_embbeding_L
        return None

    def fuse(self, img_bev_embed, pts_bev_embed):
        """channel_feature_norm -> spatial_feature_norm -> multi_modal_fusion -> permute
        (transformer_fusion.py:535-549) as one kernel.  Returns (Nq, bs, C*s).

        The variants no shipped config selects keep that kernel for the weighting, the sum / concatenation and the
        permute; what they add in front of it (a token-axis Linear, the modality projections) or behind it (the
        modal embedding) runs as device-side framework ops around the MFMA Linear."""
        ref = img_bev_embed if img_bev_embed is not None else pts_bev_embed
        cw_img, cw_pts = self._channel_factors(ref.device)
        sw_img, sw_pts = self._spatial_factors()
        cat = self.fusion_method == 'cat'
        if self.feature_norm in _MLP_NORMS or self.feature_norm == 'ModalityProjection':
            # (a missing modality is a zero map in these variants, transformer_fusion.py:318-321)
            img = img_bev_embed if img_bev_embed is not None else torch.zeros_like(ref)
            pts = pts_bev_embed if pts_bev_embed is not None else torch.zeros_like(ref)
        if self.feature_norm in _MLP_NORMS:
            iw, pw = self._learned_channel_weights(img, pts)
            fused = UF.bev_fuse(img * iw[:, None, :], pts * pw[:, None, :], cw_img, cw_pts, sw_img, sw_pts, cat=cat)
        elif self.feature_norm == 'ModalityProjection':
            # [img | proj(img)] * [c | 1 - l] + [proj(pts) | pts] * [1 - c | l]  (:287-300): each half is one
            # weighted sum of a real and a projected map
            pseudo_pts = self.l_modal_proj(img)
            pseudo_img = self.c_modal_proj(pts)
            one = torch.ones(self.embed_dims, dtype=torch.float32, device=ref.device)
            c, l = float(self.c_flag), float(self.l_flag)
            first = UF.bev_fuse(img, pseudo_img, one * c, one * (1.0 - c), sw_img, sw_pts, cat=False)
            second = UF.bev_fuse(pseudo_pts, pts, one * (1.0 - l), one * l, sw_img, sw_pts, cat=False)
            fused = torch.cat((first, second), -1)
        else:
            fused = UF.bev_fuse(img_bev_embed, pts_bev_embed, cw_img, cw_pts, sw_img, sw_pts, cat=cat)
        emb = self._modal_embedding(ref)
        if emb is not None:
            fused = fused + emb.to(fused.dtype)
        return fused

    # -- forward ---------------------------------------------------------------------------------
    def sample_modality_flags(self, has_img=True, has_pts=True):
        """(c_flag, l_flag) of one forward pass: the modality-dropout draw of
        transformer_fusion.py:463-477 (two ``np.random`` draws at most, in the reference's order)
        followed by the missing-modality rule (:482-489).  Does not touch the module."""
        l_flag = c_flag = 1
        if self.drop_modality is not None and self.training is True:
            if isinstance(self.drop_modality, dict):
                dropout_prob = self.drop_modality['dropout_prob']
                lidar_prob = self.drop_modality['lidar_prob']
            elif isinstance(self.drop_modality, float):
                dropout_prob = lidar_prob = self.drop_modality
            else:
                raise ValueError('Unrecognized type: {}'.format(type(self.drop_modality)))
            if self.get_probability(dropout_prob):
                l_flag = self.get_probability(lidar_prob) * 1
                c_flag = 1 - l_flag
        if not has_img:
            c_flag = 0
        elif not has_pts:
            l_flag = 0
        return c_flag, l_flag

    def _draw_modality_flags(self, img_mlvl_feats, pts_mlvl_feats):
        # ``forced_flags``: the caller drew (graph_step.GraphedStep replays one captured graph per
        # flag combination and draws on the host with ``sample_modality_flags``)
        forced = getattr(self, 'forced_flags', None)
        self.c_flag, self.l_flag = forced if forced is not None else self.sample_modality_flags(
            img_mlvl_feats is not None, pts_mlvl_feats is not None)
        ref = img_mlvl_feats if img_mlvl_feats is not None else pts_mlvl_feats
        return ref[0].size(0)

    def encode(self, img_mlvl_feats, pts_mlvl_feats, bev_queries, bev_h, bev_w, bev_pos=None,
               return_parts=False, **kwargs):
        """The hot path: everything of ``forward`` up to ``fused_bev_embed`` (Nq, bs, C*s)."""
        with lowp_step_cache():
            return self._encode(img_mlvl_feats, pts_mlvl_feats, bev_queries, bev_h, bev_w, bev_pos,
                                return_parts, **kwargs)

    def _encode(self, img_mlvl_feats, pts_mlvl_feats, bev_queries, bev_h, bev_w, bev_pos,
                return_parts, **kwargs):
        bs = self._draw_modality_flags(img_mlvl_feats, pts_mlvl_feats)
        pos_base = None
        if bev_pos is not None:
            base = getattr(bev_pos, '_ubv_pos_hw', None)            # (h, w, C) table behind the batch-expanded bev_pos
            if base is not None and base.is_cuda and base.dtype == torch.float32 and \
                    not torch.is_autocast_enabled('cuda') and base.shape[0] * base.shape[1] == bev_h * bev_w:
                pos_base = base.reshape(bev_h * bev_w, -1)
            bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        if torch.is_autocast_enabled('cuda') and \
                (bev_queries[0] if isinstance(bev_queries, list) else bev_queries).is_cuda:
            # Residual stream under autocast.  lowp_stream (default): 16-bit, as the reference's own
            # mixed-precision mode runs it (mmcv wrap_fp16_model: activations in half, norm
            # statistics in f32) — the queries enter in the autocast dtype, every fused
            # add+dropout+LayerNorm returns it, and no per-Linear cast / f32 gradient accumulation
            # pass remains.  Otherwise f32 (torch.autocast's LayerNorm convention).
            adt = torch.get_autocast_dtype('cuda') if self.lowp_stream else torch.float32
            bev_queries = [q.to(adt) for q in bev_queries] if isinstance(bev_queries, list) \
                else bev_queries.to(adt)
            # (hw, bs, C) view of a batch-expanded tensor: cast one sample, keep the expansion
            bev_pos = None if bev_pos is None else \
                cast_keep_expand(bev_pos.permute(1, 0, 2), adt).permute(1, 0, 2)
        two = img_mlvl_feats is not None and pts_mlvl_feats is not None and _TWO_STREAMS[0] and \
            (bev_queries[0] if isinstance(bev_queries, list) else bev_queries).is_cuda
        pos_img = pos_pts = pos_base
        if self.dual_queries:
            assert isinstance(bev_queries, list)
            tab_img, tab_pts = bev_queries[0], bev_queries[1]
        else:
            tab_img = tab_pts = bev_queries
            if two and torch.is_grad_enabled() and bev_queries.requires_grad and bev_queries.dtype == torch.float32:
                # both encoders read ONE query table (and one positional table): each gets its own alias, and the two
                # gradients meet in functional.fan_out's backward — this library's add — instead of the autograd engine's
                # framework add at the join of the two streams (unibev_amd/debug.py)
                tab_img, tab_pts = UF.fan_out(bev_queries)
        if two and torch.is_grad_enabled() and pos_base is not None and pos_base.requires_grad:
            pos_img, pos_pts = UF.fan_out(pos_base)
        q_img = tab_img.unsqueeze(1).expand(-1, bs, -1)
        q_pts = q_img if tab_pts is tab_img else tab_pts.unsqueeze(1).expand(-1, bs, -1)
        # the encoders' first layer computes its self-attention once for the batch and takes the table
        # itself for that: the attribute rides on the expanded view (a Python attribute, not part of the graph)
        if _QUERY_TABLE:                                    # (UBV_QUERY_TABLE=0: A/B runs)
            q_img._ubv_table = tab_img
            if q_pts is not q_img:
                q_pts._ubv_table = tab_pts
        img_bev_embed = pts_bev_embed = None

        l2i = None
        if img_mlvl_feats is not None and kwargs.get('img_metas') is not None and q_img.is_cuda:
            from .encoders import _lidar2img_tensor
            l2i = _lidar2img_tensor(kwargs['img_metas'], q_img.device)     # (ahead of the fork: may stack / upload)

        def run_img():
            flat, ss, lsi = self._pre_process_img_feats(img_mlvl_feats, q_img)
            kw = dict(kwargs, lidar2img_tensor=l2i) if l2i is not None else kwargs
            return self.img_bev_encoder(q_img, flat, flat, bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos,
                                        spatial_shapes=ss, level_start_index=lsi, bev_pos_base=pos_img, **kw)

        def run_pts():
            flat, ss, lsi = self._pre_process_pts_feats(pts_mlvl_feats, q_pts)
            return self.pts_bev_encoder(q_pts, flat, flat, bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos,
                                        spatial_shapes=ss, level_start_index=lsi, bev_pos_base=pos_pts, **kwargs)

        ref_q = q_img if img_mlvl_feats is not None else q_pts
        if two:
            # The two encoders are independent until the fusion: two HIP streams, forked
            # from and joined into the caller's (autograd replays every backward op on its forward
            # op's stream, so the backward forks the same way).  Most kernels of the path leave part
            # of the chip idle — tails of 1.2 - 1.6 block rounds, latency-bound GEMM phases next to
            # issue-bound sampling kernels — and the other encoder's kernels fill it.
            dev = ref_q.device
            cur = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            # the image encoder stays on the caller's stream, the point-cloud encoder gets the side stream: two
            # active streams (three — caller idle + one per encoder — measured the same, and a fourth stream of
            # any kind, e.g. RCCL's, then falls back to the single-stream time; round 5, tools/ab/job_r5p1.sh: the
            # encoders swapped between the streams 149.8 -> 148.9 samples/s, a high-priority side stream -> 115)
            hook = _REGION_HOOK[0]                  # debug.ForeignKernelLog: where the two-stream windows begin and end
            if hook is not None:
                hook('fork')
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                pts_bev_embed = run_pts()
            img_bev_embed = run_img()
            cur.wait_stream(side)
            pts_bev_embed.record_stream(cur)
            if hook is not None:
                hook('join')
                # the backward's window: from the first gradient that enters an encoder to the last one that leaves
                for t in (img_bev_embed, pts_bev_embed):
                    if t.requires_grad:
                        t.register_hook(lambda g, h=hook: h('bwd_fork'))
                ins = list(bev_queries) if isinstance(bev_queries, list) else [bev_queries]
                ins += list(img_mlvl_feats) + list(pts_mlvl_feats) + ([bev_pos] if bev_pos is not None else [])
                for t in ins:
                    if isinstance(t, torch.Tensor) and t.requires_grad:
                        t.register_hook(lambda g, h=hook: h('bwd_mark'))
        else:
            if img_mlvl_feats is not None:
                img_bev_embed = run_img()
            if pts_mlvl_feats is not None:
                pts_bev_embed = run_pts()
        fused = self.fuse(img_bev_embed, pts_bev_embed)
        if return_parts:
            return fused, img_bev_embed, pts_bev_embed
        return fused

    def forward(self, img_mlvl_feats, pts_mlvl_feats, bev_queries, object_query_embed, bev_h,
                bev_w, bev_pos=None, reg_branches=None, cls_branches=None, **kwargs):
        """-> (fused_bev_embed (Nq, bs, C*s), inter_states, init_reference_out,
        inter_references_out); transformer_fusion.py:416-586."""
        kwargs.pop('grid_length', None) if self.decoder is None else None
        fused_bev_embed = self.encode(img_mlvl_feats, pts_mlvl_feats, bev_queries, bev_h, bev_w,
                                      bev_pos=bev_pos, **kwargs)
        bs = fused_bev_embed.size(1)
        query_pos, query = torch.split(object_query_embed, self.embed_dims * self.scale_factor,
                                       dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        init_reference_out = reference_points
        if self.decoder is None:
            return fused_bev_embed, None, init_reference_out, None
        query = query.permute(1, 0, 2)
        query_pos = query_pos.permute(1, 0, 2)
        inter_states, inter_references = self.decoder(
            query=query, key=None, value=fused_bev_embed, query_pos=query_pos,
            reference_points=reference_points, reg_branches=reg_branches,
            cls_branches=cls_branches,
            spatial_shapes=shapes_tensor([(bev_h, bev_w)], query.device),
            level_start_index=index_tensor([0], query.device), **kwargs)
        return fused_bev_embed, inter_states, init_reference_out, inter_references
