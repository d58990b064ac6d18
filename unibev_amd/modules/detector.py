"""``UniBEV`` — the detector the shipped configs name as ``model.type`` (registry key
``DETECTORS:'UniBEV'``).

Reference: models/detectors/unibev_detector.py:17 (registration), :31-84 (constructor; the sub-module construction is
[ext] mmdet3d ``MVXTwoStageDetector.__init__``: ``pts_voxel_layer`` -> ``Voxelization(**cfg)``, voxel / middle encoder,
backbones, necks and ``pts_bbox_head`` through their registries, ``train_cfg.pts`` / ``test_cfg.pts`` handed to the
head), :86-110 ``extract_img_feat``, :111-124 ``extract_pts_feat``, :141-148 ``extract_feat``, :151-175 ``voxelize``,
:224-294 ``forward`` / ``forward_train``, :296-345 ``forward_test`` / ``simple_test``.

Everything the class ties together runs through the HIP library (``Voxelization``, ``HardSimpleVFE``,
``SparseEncoder``, DCNv2, ``GridMask``, the BEV encoder); plain convolutions and batch norms are MIOpen through
torch.nn.  The Hungarian loss and the NMS-free box decoding are out of scope: ``forward_train``
and ``simple_test`` run the whole forward and then raise from ``pts_bbox_head.loss`` / ``get_bboxes`` — use
``forward_outs`` (head outputs) or ``forward_bev`` (``fused_bev_embed`` only, the hot path) instead.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import BACKBONES, DETECTORS, HEADS, MIDDLE_ENCODERS, NECKS, VOXEL_ENCODERS
from .grid_mask import GridMask
from .voxel import Voxelization, voxelize_cat


def _build(cfg, registry):
    return registry.build(dict(cfg)) if cfg else None


@DETECTORS.register_module()
class UniBEV(nn.Module):
    def __init__(self, use_lidar=True, use_camera=True, use_radar=False, use_grid_mask=False, pts_voxel_layer=None,
                 pts_voxel_encoder=None, pts_middle_encoder=None, pts_fusion_layer=None, img_backbone=None,
                 pts_backbone=None, img_neck=None, pts_neck=None, pts_bbox_head=None, img_roi_head=None,
                 img_rpn_head=None, radar_voxel_layer=None, radar_voxel_encoder=None, radar_middle_encoder=None,
                 train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        if pts_fusion_layer or img_roi_head or img_rpn_head:
            raise NotImplementedError('UniBEV: pts_fusion_layer / img_roi_head / img_rpn_head are not used by any '
                                      'shipped config and are not built')
        self.use_lidar, self.use_camera, self.use_radar = use_lidar, use_camera, use_radar
        if pts_voxel_layer:
            self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
        if pts_voxel_encoder:
            self.pts_voxel_encoder = _build(pts_voxel_encoder, VOXEL_ENCODERS)
        if pts_middle_encoder:
            self.pts_middle_encoder = _build(pts_middle_encoder, MIDDLE_ENCODERS)
        if pts_backbone:
            self.pts_backbone = _build(pts_backbone, BACKBONES)
        if pts_neck is not None:
            self.pts_neck = _build(pts_neck, NECKS)
        if pts_bbox_head:
            head = dict(pts_bbox_head)
            head.update(train_cfg=train_cfg.get('pts') if train_cfg else None,
                        test_cfg=test_cfg.get('pts') if test_cfg else None)
            self.pts_bbox_head = HEADS.build(head)
        if img_backbone:
            self.img_backbone = _build(img_backbone, BACKBONES)
        if img_neck is not None:
            self.img_neck = _build(img_neck, NECKS)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.use_grid_mask = use_grid_mask
        if self.use_grid_mask:
            self.grid_mask = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)
        if radar_voxel_layer:
            self.radar_voxel_layer = Voxelization(**radar_voxel_layer)
        if radar_voxel_encoder:
            self.radar_voxel_encoder = _build(radar_voxel_encoder, VOXEL_ENCODERS)
        if radar_middle_encoder:
            self.radar_middle_encoder = _build(radar_middle_encoder, MIDDLE_ENCODERS)
        self.fusion_method = pts_bbox_head['transformer'].get('fusion_method', None) if pts_bbox_head else None
        self.fp16_enabled = False

    # [ext] MVXTwoStageDetector's ``with_*`` properties
    def _has(self, name):
        return getattr(self, name, None) is not None

    with_pts_bbox = property(lambda self: self._has('pts_bbox_head'))
    with_img_backbone = property(lambda self: self._has('img_backbone'))
    with_pts_backbone = property(lambda self: self._has('pts_backbone'))
    with_img_neck = property(lambda self: self._has('img_neck'))
    with_pts_neck = property(lambda self: self._has('pts_neck'))
    with_voxel_encoder = property(lambda self: self._has('pts_voxel_encoder'))
    with_middle_encoder = property(lambda self: self._has('pts_middle_encoder'))

    def init_weights(self):
        if self.with_pts_bbox:
            self.pts_bbox_head.init_weights()

    # ------------------------------------------------------------------------------------------- features
    def extract_img_feat(self, img, img_metas=None):
        """unibev_detector.py:86-110: (B, N, 3, H, W) images -> list of (B, N, C, h, w) per level."""
        if img is None:
            return None
        B = img.size(0)
        if img.dim() == 5 and img.size(0) == 1:
            img = img.squeeze(0)             # the reference squeezes in place (``img.squeeze_()``)
        elif img.dim() == 5 and img.size(0) > 1:
            B, N, C, H, W = img.size()
            img = img.reshape(B * N, C, H, W)
        if self.use_grid_mask:
            img = self.grid_mask(img)
        img_feats = self.img_backbone(img)
        if isinstance(img_feats, dict):
            img_feats = list(img_feats.values())
        if self.with_img_neck:
            img_feats = self.img_neck(img_feats)
        out = []
        for f in img_feats:
            BN, C, H, W = f.size()
            out.append(f.view(B, int(BN / B), C, H, W))
        return out

    def _voxels_to_feats(self, voxelize, voxel_encoder, middle_encoder, pts):
        voxels, num_points, coors = voxelize(pts)
        voxel_features = voxel_encoder(voxels, num_points, coors)
        # the reference reads ``coors[-1, 0] + 1`` back from the device; the list length is the same number
        return middle_encoder(voxel_features, coors, len(pts))

    def extract_pts_feat(self, pts):
        """unibev_detector.py:111-124: list of (N_i, F) clouds -> list of (B, C, 180, 180) maps."""
        if not self.with_pts_backbone:
            return None
        x = self._voxels_to_feats(self.voxelize, self.pts_voxel_encoder, self.pts_middle_encoder, pts)
        x = self.pts_backbone(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
        return x

    def extract_radar_feat(self, radar, img_metas=None):
        """unibev_detector.py:126-139."""
        x = self._voxels_to_feats(self.radar_voxelize, self.radar_voxel_encoder, self.radar_middle_encoder, radar)
        if not self.with_pts_backbone:
            return [x]
        x = self.pts_backbone(x)
        return self.pts_neck(x) if self.with_pts_neck else x

    def extract_feat(self, img, points, radar_points=None, img_metas=None):
        """unibev_detector.py:141-148 -> (img_feats, pts_feats, radar_feats)."""
        img_feats = self.extract_img_feat(img, img_metas) if self.use_camera else None
        pts_feats = self.extract_pts_feat(points) if self.use_lidar else None
        radar_feats = self.extract_radar_feat(radar_points, img_metas) if self.use_radar else None
        return img_feats, pts_feats, radar_feats

    @staticmethod
    def _voxelize(layer, points):
        # the reference loops over the samples (one Voxelization call and one host read each); here the batch is one
        # launch chain and one read of the B voxel counts (modules/voxel.py::voxelize_cat), same result bit for bit
        return voxelize_cat(layer, points)

    @torch.no_grad()
    def voxelize(self, points):
        """unibev_detector.py:151-175 -> (voxels (sum M, T, F), num_points (sum M,), coors (sum M, 4) [b, z, y, x])."""
        return self._voxelize(self.pts_voxel_layer, points)

    @torch.no_grad()
    def radar_voxelize(self, points):
        """unibev_detector.py:177-202 (``force_fp32``: clouds are taken as f32)."""
        return self._voxelize(self.radar_voxel_layer, [p.float() for p in points])

    # -------------------------------------------------------------------------------------------- forward
    def _select_pts_feats(self, lidar_feats, radar_feats):
        if self.use_lidar and not self.use_radar:
            return lidar_feats
        if self.use_radar and not self.use_lidar:
            return radar_feats
        if self.use_lidar and self.use_radar:
            raise ValueError('Unsupported Modality Mode: Cam: {}, Lidar:{}, Radar:{}'.format(
                self.use_camera, self.use_lidar, self.use_radar))
        return None

    def _feats(self, points, img_metas, img, radar):
        if self.use_camera:
            assert img is not None
        if self.use_lidar:
            assert points is not None
        if self.use_radar:
            assert radar is not None
        img_feats, lidar_feats, radar_feats = self.extract_feat(img=img, points=points, radar_points=radar,
                                                                img_metas=img_metas)
        return img_feats, self._select_pts_feats(lidar_feats, radar_feats)

    def forward_outs(self, points=None, img_metas=None, img=None, radar=None):
        """Backbones -> BEV encoder + fusion -> decoder -> branches: the ``outs`` dict that
        unibev_detector.py:286 / :329 hands to the loss / the box decoder."""
        img_feats, pts_feats = self._feats(points, img_metas, img, radar)
        return self.pts_bbox_head(img_feats, pts_feats, img_metas)

    def forward_bev(self, points=None, img_metas=None, img=None, radar=None):
        """Backbones -> ``fused_bev_embed`` (Nq, bs, C*s): the hot path without the object decoder."""
        img_feats, pts_feats = self._feats(points, img_metas, img, radar)
        return self.pts_bbox_head.forward_bev(img_feats, pts_feats, img_metas)

    def forward(self, return_loss=True, **kwargs):
        """unibev_detector.py:208-222."""
        return self.forward_train(**kwargs) if return_loss else self.forward_test(**kwargs)

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None,
                      gt_bboxes=None, img=None, radar=None, proposals=None, gt_bboxes_ignore=None, img_depth=None,
                      img_mask=None):
        """unibev_detector.py:224-294.  The loss is out of scope: ``pts_bbox_head.loss`` raises."""
        outs = self.forward_outs(points, img_metas, img, radar)
        losses = dict()
        losses.update(self.pts_bbox_head.loss(gt_bboxes_3d, gt_labels_3d, outs, img_metas=img_metas))
        return losses

    def forward_test(self, img_metas, img=None, points=None, radar=None, **kwargs):
        """unibev_detector.py:296-316."""
        if not isinstance(img_metas, list):
            raise TypeError('{} must be a list, but got {}'.format('img_metas', type(img_metas)))
        img = [img] if img is None else img
        points = [points] if points is None else points
        radar = [radar] if radar is None else radar
        bbox_results, _ = self.simple_test(points[0], img_metas[0], img[0], radar[0], **kwargs)
        return bbox_results

    def simple_test(self, points, img_metas, img=None, radar=None, rescale=False):
        """unibev_detector.py:318-345.  Box decoding is out of scope: ``pts_bbox_head.get_bboxes`` raises."""
        outs = self.forward_outs(points, img_metas, img, radar)
        bbox_list_head = self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale)
        bbox_list = [dict(pts_bbox=dict(boxes_3d=b, scores_3d=s, labels_3d=l)) for b, s, l in bbox_list_head]
        return bbox_list, outs['bev_embed']

    def forward_dummy(self, img, points):
        return self.forward_test(img=img, points=points, img_metas=[[None]])
