"""LiDAR front end modules: ``Voxelization`` (hard, deterministic), ``HardSimpleVFE`` and the
sparse -> dense scatter, with the [ext] mmdet3d 0.18.1 constructor / call conventions the
reference reaches from models/detectors/unibev_detector.py:112-175
(config: configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:186-193).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as UF
from ..registry import VOXEL_ENCODERS


class Voxelization(nn.Module):
    """points (N, F) -> voxels (M, T, F), coors (M, 3) int32 zyx, num_points (M,) int32.

    ``max_voxels`` is (train, test) as in mmdet3d.  ``max_num_points == -1 or max_voxels == -1``
    selects dynamic voxelization (per-point coors only).  ``forward`` slices to M like the
    reference op, which costs one device->host read of M; ``forward_padded`` is the sync-free form
    (full-capacity buffers + device-side M) used by the fused front end.
    """

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000,
                 deterministic=True):
        super().__init__()
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else (max_voxels, max_voxels)
        self.deterministic = deterministic
        pcr = torch.tensor(self.point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(self.voxel_size, dtype=torch.float32)
        grid = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid
        self.pcd_shape = [*grid.tolist(), 1][::-1]

    def _limit(self):
        return self.max_voxels[0] if self.training else self.max_voxels[1]

    def forward_padded(self, points):
        return UF.hard_voxelize(points, self.voxel_size, self.point_cloud_range,
                                self.max_num_points, self._limit())

    def forward(self, points):
        if self.max_num_points == -1 or self._limit() == -1:
            return UF.dynamic_voxelize(points, self.voxel_size, self.point_cloud_range)
        voxels, coors, num, vnum = self.forward_padded(points)
        m = int(vnum.item())
        return voxels[:m], coors[:m], num[:m]

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range='
                f'{self.point_cloud_range}, max_num_points={self.max_num_points}, max_voxels='
                f'{self.max_voxels}, deterministic={self.deterministic})')


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    """Mean of the points of each voxel: (M, T, F) -> (M, num_features)."""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors=None):
        mean = UF.voxel_mean(features, num_points)
        return mean[:, :self.num_features].contiguous()


def voxelize_batch(voxel_layer, points):
    """``UniBEV.voxelize`` (unibev_detector.py:151-175): per-sample voxelization, concatenation and
    the batch index prepended to coors -> (voxels, num_points, coors_batch (sum M, 4))."""
    voxels, coors, num_points = [], [], []
    for i, res in enumerate(points):
        v, c, n = voxel_layer(res)
        voxels.append(v)
        coors.append(F.pad(c, (1, 0), mode='constant', value=i))
        num_points.append(n)
    return torch.cat(voxels, 0), torch.cat(num_points, 0), torch.cat(coors, 0)


def sparse_to_dense(features, coors, batch_size, spatial_shape):
    """``SparseConvTensor.dense()`` + the reference's (N, C*D, H, W) view (SparseEncoder tail)."""
    dense = UF.sparse_to_dense(features, coors.int(), batch_size, spatial_shape)
    N, C, D, H, W = dense.shape
    return dense.view(N, C * D, H, W)
