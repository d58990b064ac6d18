"""CPU oracle of GridMask (the device-side augmentation) — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/projects/UniBEV/unibev_plugin/models/utils/grid_mask.py:70-123 (``GridMask.forward``) as the
detector builds it (models/detectors/unibev_detector.py:75: use_h = use_w = True, rotate = 1, offset = False,
ratio = 0.5, mode = 1, prob = 0.7).  Pinned by tests/golden/grid_mask.npz (recorded from the reference).

Draw order on ``np.random`` (grid_mask.py:94-113): rand() for the skip test; then randint(2, h) -> d,
randint(d) -> st_h, randint(d) -> st_w, randint(rotate) -> r, and with ``offset`` h * w draws of rand() for the fill.
With rotate = 1, r is always 0 and the PIL rotation is the identity.  For r > 0 the reference calls
``PIL.Image.rotate(r)`` on the uint8 grid ([ext] Pillow, not vendored): ``rotate_nearest`` below restates Pillow's
published algorithm for that call (Image.rotate: the 0 / 180 / square 90 / 270 shortcuts, otherwise an affine
transform about the image centre with the matrix entries rounded to 15 decimals; Geometry.c affine_fixed: 16.16
fixed-point nearest-neighbour sampling at pixel centres, zero fill) and is pinned by the rotated masks recorded from
the reference (tests/golden/grid_mask.npz, ``v*`` entries).
"""
import math

import numpy as np


def draw(h, prob, rotate=1, ratio=0.5, training=True, rng=np.random):
    """The reference's random draws, in its order: None when the pass is skipped, else (d, l, st_h, st_w, r)."""
    if rng.rand() > prob or not training:
        return None
    d = rng.randint(2, h)
    length = min(max(int(d * ratio + 0.5), 1), d - 1)
    st_h = rng.randint(d)
    st_w = rng.randint(d)
    r = rng.randint(rotate)
    return d, length, st_h, st_w, r


def rotate_nearest(img, angle):
    """``np.asarray(PIL.Image.fromarray(img).rotate(angle))`` for a 2-D uint8 array (nearest neighbour, same size,
    zero fill) — Pillow's Image.rotate + Geometry.c::affine_fixed, see the module docstring."""
    hgt, wid = img.shape
    angle = angle % 360.0
    if angle == 0:
        return img.copy()
    if angle == 180:
        return img[::-1, ::-1].copy()
    if angle in (90, 270) and wid == hgt:
        return np.rot90(img, 1 if angle == 90 else 3).copy()          # counter-clockwise
    rad = -math.radians(angle)
    a = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    cx, cy = wid / 2.0, hgt / 2.0
    a[2] = a[0] * -cx + a[1] * -cy + a[2] + cx
    a[5] = a[3] * -cx + a[4] * -cy + a[5] + cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(v) if v >= 0.0 else int(math.floor(v))
    a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
    a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
    a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    ys, xs = np.meshgrid(np.arange(hgt, dtype=np.int64), np.arange(wid, dtype=np.int64), indexing='ij')
    xin = (a2 + a1 * ys + a0 * xs) >> 16
    yin = (a5 + a4 * ys + a3 * xs) >> 16
    ok = (xin >= 0) & (xin < wid) & (yin >= 0) & (yin < hgt)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def fill_draw(h, w, rng=np.random):
    """The ``offset`` fill of the masked pixels (grid_mask.py:120): h * w draws, uniform in [-1, 1)."""
    return (2 * (rng.rand(h, w) - 0.5)).astype(np.float32)


def mask(h, w, d, length, st_h, st_w, r=0, use_h=True, use_w=True, mode=1):
    """(h, w) float32 multiplier (grid_mask.py:97-118)."""
    hh, ww = int(1.5 * h), int(1.5 * w)
    m = np.ones((hh, ww), np.float32)
    if use_h:
        for i in range(hh // d):
            s = d * i + st_h
            m[s:min(s + length, hh), :] = 0
    if use_w:
        for i in range(ww // d):
            s = d * i + st_w
            m[:, s:min(s + length, ww)] = 0
    if r != 0:
        m = rotate_nearest(m.astype(np.uint8), r).astype(np.float32)
    m = m[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w]
    return 1 - m if mode == 1 else m
