"""``GridMask``: the detector's image-side augmentation, evaluated on the device.

Reference: models/utils/grid_mask.py:70-123, built at models/detectors/unibev_detector.py:75 as
``GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)`` and applied to the (B*N, C, H, W) image
batch in ``extract_img_feat``.  Same constructor, same ``set_prob``, same draws from ``np.random`` in the same order
(a seeded run masks the same pixels); the mask itself is never built — ``ubv_grid_mask`` evaluates its closed form
while multiplying.  ``rotate > 1`` (a PIL rotation of the grid) and ``offset=True`` are not supported: no shipped
config uses them.
"""
import numpy as np
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import functional as UF


class _GridMaskFn(Function):
    @staticmethod
    def forward(ctx, x, geom):
        ctx.geom = geom
        return UF.grid_mask(x, *geom)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return UF.grid_mask(g, *ctx.geom), None


class GridMask(nn.Module):
    def __init__(self, use_h, use_w, rotate=1, offset=False, ratio=0.5, mode=0, prob=1.):
        super().__init__()
        if rotate != 1 or offset:
            raise NotImplementedError('GridMask: rotate > 1 and offset=True are not built (unused by the configs)')
        self.use_h, self.use_w, self.rotate, self.offset = use_h, use_w, rotate, offset
        self.ratio, self.mode, self.st_prob, self.prob = ratio, mode, prob, prob
        self.fp16_enable = False

    def set_prob(self, epoch, max_epoch):
        self.prob = self.st_prob * epoch / max_epoch

    def forward(self, x):
        if np.random.rand() > self.prob or not self.training:
            return x
        n, c, h, w = x.size()
        d = np.random.randint(2, h)
        length = min(max(int(d * self.ratio + 0.5), 1), d - 1)
        st_h = np.random.randint(d)
        st_w = np.random.randint(d)
        np.random.randint(self.rotate)                    # the reference's rotation draw (always 0 here)
        geom = (int(d), int(length), int(st_h), int(st_w), bool(self.use_h), bool(self.use_w), int(self.mode))
        return _GridMaskFn.apply(x, geom)
