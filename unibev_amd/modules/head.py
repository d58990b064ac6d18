"""``UniBEV_Head`` — the caller of the hot path (reference: models/dense_heads/unibev_head.py:26-242).

Owns what the BEV encoder consumes: ``bev_embedding`` (Nq x C BEV queries, :126-133),
``positional_encoding`` (LearnedPositionalEncoding, :179-182), ``query_embedding`` (:134-135) and
the ``transformer``; parameter names are the checkpoint keys.  The
Hungarian loss / box decoding of the DETR3D head are out of scope;
classification / regression branches are built and applied when the transformer has a decoder.
"""
import copy
import math

import torch
import torch.nn as nn

from ..registry import HEADS, build_positional_encoding, build_transformer
from .bricks import BaseModule, cast_keep_expand
from .decoder import inverse_sigmoid


@HEADS.register_module()
class UniBEV_Head(BaseModule):
    def __init__(self, *args, num_classes=10, in_channels=256, num_query=900, with_box_refine=False,
                 as_two_stage=False, transformer=None, bbox_coder=None, positional_encoding=None,
                 num_cls_fcs=2, code_weights=None, bev_h=30, bev_w=30, num_reg_fcs=2,
                 code_size=10, sync_cls_avg_factor=False, loss_cls=None, loss_bbox=None,
                 loss_iou=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        self.bev_h, self.bev_w = bev_h, bev_w
        self.num_query = num_query
        self.num_classes = num_classes
        # DETRHead: sigmoid classification (every shipped config: FocalLoss, use_sigmoid=True) has
        # num_classes outputs, softmax one more for the background.  Defaults as in [ext] mmdet: DETRHead's default
        # loss is CrossEntropyLoss (use_sigmoid=False); FocalLoss defaults to use_sigmoid=True, every other loss False
        loss_cls = dict(loss_cls) if loss_cls else dict(type='CrossEntropyLoss', use_sigmoid=False)
        self.use_sigmoid_cls = bool(loss_cls.get('use_sigmoid', loss_cls.get('type') == 'FocalLoss'))
        self.cls_out_channels = num_classes if self.use_sigmoid_cls else num_classes + 1
        self.with_box_refine = with_box_refine
        self.as_two_stage = as_two_stage
        self.code_size = code_size
        self.num_reg_fcs = num_reg_fcs
        self.num_cls_fcs = num_cls_fcs - 1
        self.fp16_enabled = False
        if self.as_two_stage:
            transformer = dict(transformer, as_two_stage=True)
        self.code_weights = nn.Parameter(torch.tensor(
            code_weights if code_weights is not None
            else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2][:code_size]), requires_grad=False)
        self.pc_range = (bbox_coder or {}).get('pc_range', [-54, -54, -5, 54, 54, 3])
        self.real_w = self.pc_range[3] - self.pc_range[0]
        self.real_h = self.pc_range[4] - self.pc_range[1]
        self.positional_encoding = build_positional_encoding(dict(positional_encoding))
        self.transformer = build_transformer(dict(transformer))
        self.embed_dims = self.transformer.embed_dims
        self.scale_factor = self.transformer.scale_factor
        self.dual_queries = self.transformer.dual_queries
        self._init_layers()

    def _init_layers(self):
        """unibev_head.py:86-135."""
        dims = self.embed_dims * self.scale_factor
        decoder = getattr(self.transformer, 'decoder', None)
        if decoder is not None:
            cls_branch = []
            for _ in range(self.num_reg_fcs):
                cls_branch += [nn.Linear(dims, dims), nn.LayerNorm(dims), nn.ReLU(inplace=True)]
            cls_branch.append(nn.Linear(dims, self.cls_out_channels))
            fc_cls = nn.Sequential(*cls_branch)
            reg_branch = []
            for _ in range(self.num_reg_fcs):
                reg_branch += [nn.Linear(dims, dims), nn.ReLU()]
            reg_branch.append(nn.Linear(dims, self.code_size))
            reg_branch = nn.Sequential(*reg_branch)
            num_pred = decoder.num_layers + 1 if self.as_two_stage else decoder.num_layers
            if self.with_box_refine:
                self.cls_branches = nn.ModuleList([copy.deepcopy(fc_cls) for _ in range(num_pred)])
                self.reg_branches = nn.ModuleList([copy.deepcopy(reg_branch) for _ in range(num_pred)])
            else:
                self.cls_branches = nn.ModuleList([fc_cls for _ in range(num_pred)])
                self.reg_branches = nn.ModuleList([reg_branch for _ in range(num_pred)])
        if not self.as_two_stage:
            if self.dual_queries:
                self.bev_embedding_img = nn.Embedding(self.bev_h * self.bev_w, self.embed_dims)
                self.bev_embedding_pts = nn.Embedding(self.bev_h * self.bev_w, self.embed_dims)
            else:
                self.bev_embedding = nn.Embedding(self.bev_h * self.bev_w, self.embed_dims)
            self.query_embedding = nn.Embedding(self.num_query, dims * 2)

    def init_weights(self):
        """unibev_head.py:137-143: the transformer's own initialisation, then the focal-loss prior
        (bias_init_with_prob(0.01)) on the last Linear of every classification branch.  Nothing
        else is touched: ``positional_encoding`` keeps nn.Embedding's N(0, 1) (the reference head's
        override never recurses into it)."""
        self.transformer.init_weights()
        if self.use_sigmoid_cls and hasattr(self, 'cls_branches'):
            bias_init = float(-math.log((1 - 0.01) / 0.01))
            for m in self.cls_branches:
                nn.init.constant_(m[-1].bias, bias_init)

    def loss(self, *args, **kwargs):
        """unibev_head.py:322-427 (Hungarian assignment + focal / L1 losses): out of scope, SURVEY.md section 2 #13."""
        raise NotImplementedError('UniBEV_Head.loss: the Hungarian loss is outside the hot path (SURVEY.md section 2); '
                                  'take the head outputs from forward() and apply your own criterion')

    def get_bboxes(self, *args, **kwargs):
        """unibev_head.py:429-456 (NMSFreeCoder decode): out of scope, SURVEY.md section 2 #16."""
        raise NotImplementedError('UniBEV_Head.get_bboxes: NMS-free box decoding is outside the hot path '
                                  '(SURVEY.md section 2); decode all_cls_scores / all_bbox_preds of forward()')

    def bev_inputs(self, bs, dtype, device):
        """(bev_queries, bev_pos) exactly as unibev_head.py:171-182 builds them."""
        if self.dual_queries:
            bev_queries = [self.bev_embedding_img.weight.to(dtype), self.bev_embedding_pts.weight.to(dtype)]
        else:
            bev_queries = self.bev_embedding.weight.to(dtype)
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=device).to(dtype)
        return bev_queries, cast_keep_expand(self.positional_encoding(bev_mask), dtype)

    def forward_bev(self, mlvl_img_feats, pts_feats, img_metas):
        """The hot path only: BEV features ``bev_embed`` (Nq, bs, C*s)."""
        ref = mlvl_img_feats[0] if mlvl_img_feats is not None else pts_feats[0]
        bev_queries, bev_pos = self.bev_inputs(ref.shape[0], ref.dtype, ref.device)
        return self.transformer.encode(mlvl_img_feats, pts_feats, bev_queries, self.bev_h,
                                       self.bev_w, bev_pos=bev_pos, img_metas=img_metas)

    def forward(self, mlvl_img_feats, pts_feats, img_metas):
        """unibev_head.py:145-242 -> dict(bev_embed, all_cls_scores, all_bbox_preds, ...)."""
        ref = mlvl_img_feats[0] if mlvl_img_feats is not None else pts_feats[0]
        bs, dtype = ref.shape[0], ref.dtype
        object_query_embeds = self.query_embedding.weight.to(dtype)
        bev_queries, bev_pos = self.bev_inputs(bs, dtype, ref.device)
        has_dec = getattr(self.transformer, 'decoder', None) is not None
        bev_embed, hs, init_reference, inter_references = self.transformer(
            mlvl_img_feats, pts_feats, bev_queries, object_query_embeds, self.bev_h, self.bev_w,
            grid_length=(self.real_h / self.bev_h, self.real_w / self.bev_w), bev_pos=bev_pos,
            reg_branches=self.reg_branches if (has_dec and self.with_box_refine) else None,
            cls_branches=self.cls_branches if (has_dec and self.as_two_stage) else None,
            img_metas=img_metas)
        outs = dict(bev_embed=bev_embed, all_cls_scores=None, all_bbox_preds=None,
                    enc_cls_scores=None, enc_bbox_preds=None)
        if not has_dec:
            return outs
        hs = hs.permute(0, 2, 1, 3)
        classes, coords = [], []
        for lvl in range(hs.shape[0]):
            reference = inverse_sigmoid(init_reference if lvl == 0 else inter_references[lvl - 1])
            outputs_class = self.cls_branches[lvl](hs[lvl])
            tmp = self.reg_branches[lvl](hs[lvl])
            assert reference.shape[-1] == 3
            xy = (tmp[..., 0:2] + reference[..., 0:2]).sigmoid()
            z = (tmp[..., 4:5] + reference[..., 2:3]).sigmoid()
            pc = self.pc_range
            tmp = torch.cat((xy[..., 0:1] * (pc[3] - pc[0]) + pc[0],
                             xy[..., 1:2] * (pc[4] - pc[1]) + pc[1], tmp[..., 2:4],
                             z * (pc[5] - pc[2]) + pc[2], tmp[..., 5:]), -1)
            classes.append(outputs_class)
            coords.append(tmp)
        outs['all_cls_scores'] = torch.stack(classes)
        outs['all_bbox_preds'] = torch.stack(coords)
        return outs
