"""``GridMask``: the detector's image-side augmentation, evaluated on the device.

Reference: models/utils/grid_mask.py:70-123, built at models/detectors/unibev_detector.py:75 as
``GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)`` and applied to the (B*N, C, H, W) image
batch in ``extract_img_feat``.  Same constructor, same ``set_prob``, same draws from ``np.random`` in the same order
(a seeded run masks the same pixels); the mask itself is never built — ``ubv_grid_mask`` evaluates its closed form
while multiplying.

``rotate > 1`` and ``offset=True`` (no shipped config uses them) are host-side constructions in the reference too — a
PIL rotation of the uint8 grid by ``randint(rotate)`` degrees and h x w draws of ``np.random.rand`` for the fill of the
masked pixels (grid_mask.py:111-122): a pass that draws a non-zero angle, or fills, builds the (h, w) mask the same way
(numpy + PIL), uploads it and applies it with two device-side element-wise ops; a zero angle without fill keeps the
closed-form kernel.
"""
import numpy as np
import torch
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
        r = np.random.randint(self.rotate)                # the reference's rotation draw (0 for rotate = 1)
        if r == 0 and not self.offset:
            geom = (int(d), int(length), int(st_h), int(st_w), bool(self.use_h), bool(self.use_w), int(self.mode))
            return _GridMaskFn.apply(x, geom)
        mask = torch.from_numpy(self._host_mask(h, w, d, length, st_h, st_w, r)).to(x.dtype).to(x.device)
        if not self.offset:
            return x * mask
        fill = torch.from_numpy(2 * (np.random.rand(h, w) - 0.5)).to(x.dtype).to(x.device)
        return x * mask + fill * (1 - mask)

    def _host_mask(self, h, w, d, length, st_h, st_w, r):
        """The (h, w) 0 / 1 multiplier of a rotated grid (grid_mask.py:97-118): stripes on the 1.5x canvas, PIL's
        nearest-neighbour rotation about its centre, centre crop, complement for mode 1."""
        hh, ww = int(1.5 * h), int(1.5 * w)
        m = np.ones((hh, ww), np.uint8)
        if self.use_h:
            for s in range(st_h, d * (hh // d) + st_h, d):
                m[s:min(s + length, hh), :] = 0
        if self.use_w:
            for s in range(st_w, d * (ww // d) + st_w, d):
                m[:, s:min(s + length, ww)] = 0
        if r != 0:
            from PIL import Image                         # the reference's own dependency for this branch
            m = np.asarray(Image.fromarray(m).rotate(r))
        m = m[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w].astype(np.float32)
        return 1 - m if self.mode == 1 else m
