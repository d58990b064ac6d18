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
        # (train, test) pair, given as tuple or list (mmdet3d's _pair), or one number for both
        self.max_voxels = tuple(max_voxels) if isinstance(max_voxels, (tuple, list)) \
            else (max_voxels, max_voxels)
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

    def forward_padded(self, features, num_points, voxel_num):
        """Full-capacity buffers + the device-side voxel count (rows >= voxel_num stay zero)."""
        return UF.voxel_mean(features, num_points, voxel_num)[:, :self.num_features]


class _DynamicScatterFn(torch.autograd.Function):
    """[ext] mmdet3d ``_dynamic_scatter``: forward through ``ubv_dynamic_point_to_voxel_forward``;
    backward routes the voxel gradients back to the points (sum: copy, mean: / count, max: to the
    points that hold the maximum)."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type):
        vf, vc, mp, cnt, vnum = UF.dynamic_scatter(feats, coors, reduce_type)
        # the kernel packs each coordinate into 63 // D (at most 21) bits of a sort key: a larger coordinate cannot be
        # told from an invalid (negative) one there, so it is rejected HERE instead of being dropped silently — in the
        # same host read that fetches M (the published op returns exact shapes)
        bits = min(21, 63 // max(int(coors.shape[1]), 1))
        too_big = (coors >= (1 << bits)).any().to(vnum.dtype).view(1) if coors.numel() else vnum.new_zeros(1)
        m, _, bad = torch.cat((vnum, too_big)).tolist()
        if bad:
            raise ValueError(f'dynamic_scatter: voxel coordinates must be below 2^{bits} = {1 << bits} '
                             f'(got max {int(coors.max())})')
        vf, vc, cnt = vf[:m], vc[:m], cnt[:m]
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, vf, mp, cnt)
        ctx.mark_non_differentiable(vc)
        return vf.to(feats.dtype), vc

    @staticmethod
    def backward(ctx, grad_vf, grad_vc=None):
        """[ext] mmdet3d ``dynamic_point_to_voxel_backward``: sum copies the voxel's gradient to its points, mean
        divides by the count, max routes each (voxel, channel) gradient to ONE point — the smallest point index whose
        feature equals the maximum (the published op's ``atomicMin`` over point indices), never to every tied point."""
        feats, vf, mp, cnt = ctx.saved_tensors
        n, c = feats.shape
        if vf.shape[0] == 0 or n == 0:                   # no valid point: nothing receives a gradient
            return torch.zeros_like(feats), None, None
        valid = mp >= 0
        idx = mp.clamp(min=0).long()
        g = grad_vf[idx]
        if ctx.reduce_type == 'mean':
            g = g / cnt[idx].to(g.dtype)[:, None]
        elif ctx.reduce_type == 'max':
            pidx = torch.arange(n, device=feats.device).view(n, 1).expand(n, c)
            tied = (feats.float() == vf[idx]) & valid[:, None]
            first = torch.full((vf.shape[0], c), n, dtype=torch.long, device=feats.device)
            first.scatter_reduce_(0, idx.view(n, 1).expand(n, c), torch.where(tied, pidx, n), 'amin')
            g = g * (first[idx] == pidx).to(g.dtype)
        return (g * valid[:, None].to(g.dtype)).to(feats.dtype), None, None


def dynamic_scatter(feats, coors, reduce_type='max'):
    """(voxel_feats (M, C), voxel_coors (M, D)) as [ext] mmdet3d ``dynamic_scatter``."""
    return _DynamicScatterFn.apply(feats, coors, reduce_type)


class DynamicScatter(nn.Module):
    """[ext] mmdet3d ``DynamicScatter(voxel_size, point_cloud_range, average_points)``: reduce the
    features of the points of each voxel (mean if ``average_points`` else max).  ``coors`` is
    (N, 3) zyx for one sample or (N, 4) with the batch index first.  The published module loops
    over the batch with one host sync per sample; the batch index is simply the leading key here
    (lexicographic order = per-sample results concatenated), so a batch is one launch.
    ``forward_padded`` is the sync-free form: full-capacity outputs + device-side count."""

    def __init__(self, voxel_size, point_cloud_range, average_points):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    @property
    def reduce_type(self):
        return 'mean' if self.average_points else 'max'

    def forward_single(self, points, coors):
        return dynamic_scatter(points.contiguous(), coors.contiguous(), self.reduce_type)

    def forward_padded(self, points, coors):
        return UF.dynamic_scatter(points, coors, self.reduce_type)

    def forward(self, points, coors):
        return self.forward_single(points, coors)

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range='
                f'{self.point_cloud_range}, average_points={self.average_points})')


def voxelize_batch_padded(voxel_layer, points):
    """Sync-free ``UniBEV.voxelize``: per-sample full-capacity buffers and device-side voxel counts;
    nothing is sliced, so no voxel count is read back.  Returns lists of (voxels, coors, num, vnum)
    per sample (``HardSimpleVFE.forward_padded`` / ``sparse_to_dense`` take the device counts).  The whole batch is
    ONE launch chain (``ubv_hard_voxelize_batch``)."""
    v, c, n, m = UF.hard_voxelize_batch(list(points), voxel_layer.voxel_size, voxel_layer.point_cloud_range,
                                        voxel_layer.max_num_points, voxel_layer._limit())
    return [(v[b], c[b], n[b], m[b:b + 1]) for b in range(len(points))]


def voxelize_batch(voxel_layer, points):
    """``UniBEV.voxelize`` (unibev_detector.py:151-175): per-sample voxelization, concatenation and
    the batch index prepended to coors -> (voxels, num_points, coors_batch (sum M, 4))."""
    return voxelize_cat(voxel_layer, points)


def voxelize_cat(voxel_layer, points, with_mean=False):
    """Hard voxelization of a list of clouds in one launch chain, then the reference's concatenated form (one read of
    the B voxel counts instead of one per sample).  Dynamic voxelization keeps the per-sample calls.  ``with_mean``: a
    fourth result, the per-voxel point means (sum M, F) from the same chain (None when the chain was not used)."""
    if voxel_layer.max_num_points == -1 or voxel_layer._limit() == -1 or len(points) == 0:
        out = [voxel_layer(res) for res in points]
        coors = [F.pad(c, (1, 0), mode='constant', value=i) for i, (_, c, _) in enumerate(out)]
        res = (torch.cat([o[0] for o in out], 0), torch.cat([o[2] for o in out], 0), torch.cat(coors, 0))
        return res + (None,) if with_mean else res
    got = UF.hard_voxelize_batch(list(points), voxel_layer.voxel_size, voxel_layer.point_cloud_range,
                                 voxel_layer.max_num_points, voxel_layer._limit(), with_mean=with_mean)
    v, c, n, m = got[:4]
    counts = m.tolist()
    voxels = torch.cat([v[b, :k] for b, k in enumerate(counts)], 0)
    num_points = torch.cat([n[b, :k] for b, k in enumerate(counts)], 0)
    coors = torch.cat([F.pad(c[b, :k], (1, 0), mode='constant', value=b) for b, k in enumerate(counts)], 0)
    if with_mean:
        return voxels, num_points, coors, torch.cat([got[4][b, :k] for b, k in enumerate(counts)], 0)
    return voxels, num_points, coors


def sparse_to_dense(features, coors, batch_size, spatial_shape):
    """``SparseConvTensor.dense()`` + the reference's (N, C*D, H, W) view (SparseEncoder tail)."""
    dense = UF.sparse_to_dense(features, coors.int(), batch_size, spatial_shape)
    N, C, D, H, W = dense.shape
    return dense.view(N, C * D, H, W)


def extract_pts_feat(points, voxel_layer, voxel_encoder, middle_encoder, backbone=None, neck=None):
    """The LiDAR branch of ``UniBEV.extract_pts_feat`` (unibev_detector.py:111-123) up to — and, when given,
    through — the 2-D backbone / neck: list of per-sample point clouds -> voxelize (``voxelize_batch``) ->
    voxel encoder -> middle encoder (``SparseEncoder``: (B, 256, 180, 180) at the shipped configs).  The batch
    size is taken from the list, not read back from ``coors[-1, 0]`` as the reference does."""
    if type(voxel_encoder) is HardSimpleVFE:
        # the mean of each voxel's points comes out of the voxelization chain itself (ubv_hard_voxelize_batch_vfe)
        voxels, num_points, coors, mean = voxelize_cat(voxel_layer, points, with_mean=True)
        voxel_features = (mean[:, :voxel_encoder.num_features].contiguous() if mean is not None
                          else voxel_encoder(voxels, num_points, coors))
    else:
        voxels, num_points, coors = voxelize_batch(voxel_layer, points)
        voxel_features = voxel_encoder(voxels, num_points, coors)
    x = middle_encoder(voxel_features, coors, len(points))
    if backbone is not None:
        x = backbone(x)
        if neck is not None:
            x = neck(x)
    return x
