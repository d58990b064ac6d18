"""CPU oracle of GridMask (the device-side augmentation) — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/projects/UniBEV/unibev_plugin/models/utils/grid_mask.py:70-123 (``GridMask.forward``) as the
detector builds it (models/detectors/unibev_detector.py:75: use_h = use_w = True, rotate = 1, offset = False,
ratio = 0.5, mode = 1, prob = 0.7).  Pinned by tests/golden/grid_mask.npz (recorded from the reference).

Draw order on ``np.random`` (grid_mask.py:94-113): rand() for the skip test; then randint(2, h) -> d,
randint(d) -> st_h, randint(d) -> st_w, randint(rotate) -> r.  With rotate = 1, r is always 0 and the PIL rotation
is the identity; other rotations are outside this restatement.
"""
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


def mask(h, w, d, length, st_h, st_w, r=0, use_h=True, use_w=True, mode=1):
    """(h, w) float32 multiplier (grid_mask.py:97-118)."""
    if r != 0:
        raise NotImplementedError('rotated grids (rotate > 1)')
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
    m = m[(hh - h) // 2:(hh - h) // 2 + h, (ww - w) // 2:(ww - w) // 2 + w]
    return 1 - m if mode == 1 else m
