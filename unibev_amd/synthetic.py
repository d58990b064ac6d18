"""Synthetic nuScenes-shaped inputs (BASELINE.json north_star).

No dataset or checkpoint is reachable, so the bench, the GPU parity tests and the golden
generator all draw their inputs from here.  Everything is produced by ``numpy.random.RandomState``
(stable across numpy versions) keyed by a seed and a name, so the CPU container and the GPU box
regenerate bit-identical arrays.
"""
import math
import zlib

import numpy as np

PC_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
VOXEL_SIZE = (0.075, 0.075, 0.2)


def seeded_array(name, shape, seed=0, scale=1.0, dtype=np.float32):
    """Deterministic N(0, scale^2) array keyed by (seed, name)."""
    key = (int(seed) * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF
    rs = np.random.RandomState(key)
    return (rs.standard_normal(tuple(shape)) * scale).astype(dtype)


def param_scale(name, shape):
    """Scale used for seeded parameters: non-degenerate (unlike the reference's zero init of
    sampling_offsets / attention_weights) but well conditioned."""
    leaf = name.split('.')[-1]
    if 'sampling_offsets' in name:
        return 0.12 if leaf == 'weight' else 2.0     # offsets of a few pixels
    if 'attention_weights' in name:
        return 0.08 if leaf == 'weight' else 0.5
    if len(shape) >= 2:
        return 1.0 / math.sqrt(shape[-1])
    if 'norms' in name and leaf == 'weight':
        return 0.1                                    # added to 1 below
    if leaf in ('img_channel_weights', 'pts_channel_weights',
                'img_spatial_weights', 'pts_spatial_weights'):
        return 1.0
    return 0.1


def seeded_state_dict(named_shapes, seed=0):
    """name -> float32 ndarray for every (name, shape); LayerNorm weights centred on 1."""
    out = {}
    for name, shape in named_shapes:
        a = seeded_array('param:' + name, shape, seed, param_scale(name, tuple(shape)))
        if 'norms' in name and name.endswith('weight'):
            a = a + 1.0
        out[name] = a.astype(np.float32)
    return out


def camera_rig(num_cams=6, img_hw=(256, 704), dtype=np.float64):
    """lidar2img (num_cams,4,4) for a nuScenes-like surround rig.

    Yaws 0, -55, +55, 180, -110, +110 degrees from the +y (forward) axis; focal 557 px at 704 px
    width (back camera 356 px), principal point at the image centre, cameras 1 m from the origin
    and 0.3 m below the LiDAR.  With fewer than 6 cameras the first
    ``num_cams`` yaws are used.
    """
    h, w = img_hw
    yaws = [0.0, -55.0, 55.0, 180.0, -110.0, 110.0]
    mats = []
    for i in range(num_cams):
        yaw = math.radians(yaws[i % 6])
        f = (356.0 if yaws[i % 6] == 180.0 else 557.0) * (w / 704.0)
        fwd = np.array([math.sin(yaw), math.cos(yaw), 0.0])
        right = np.array([math.cos(yaw), -math.sin(yaw), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        c = fwd * 1.0 + np.array([0.0, 0.0, -0.3])
        R = np.stack([right, down, fwd], 0)                # cam <- lidar
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ c
        K = np.eye(4)
        K[0, 0] = K[1, 1] = f
        K[0, 2] = w / 2.0
        K[1, 2] = h / 2.0
        mats.append(K @ E)
    return np.stack(mats, 0).astype(dtype)


def img_metas(bs, num_cams=6, img_hw=(256, 704), jitter_seed=None):
    """List of per-sample meta dicts with the two keys the path reads
    (reference: encoder_unibev_detr_img.py:115-118, 166-167)."""
    metas = []
    for b in range(bs):
        l2i = camera_rig(num_cams, img_hw)
        if jitter_seed is not None and b > 0:
            # per-sample extrinsic jitter so that batch elements differ (exposes quirk q1)
            l2i = l2i + seeded_array(f'l2i_jitter{b}', l2i.shape, jitter_seed, 1e-2, np.float64) \
                * np.abs(l2i)
        metas.append(dict(lidar2img=[m for m in l2i],
                          img_shape=[(img_hw[0], img_hw[1], 3)] * num_cams))
    return metas


def lidar_points(n=30000, seed=0, frac_outside=0.02, pc_range=PC_RANGE):
    """(n,5) float32 cloud: r = 54 u^2, theta ~ U(0, 2pi), z ~ N(-1, 1) clipped to [-5, 3),
    intensity ~ U[0, 255), dt = 0, plus ``frac_outside`` points beyond the range
    (SURVEY.md section 8(d) cfg3)."""
    rs = np.random.RandomState((seed * 7919 + 17) & 0x7FFFFFFF)
    r = 54.0 * rs.random_sample(n) ** 2
    th = rs.random_sample(n) * 2.0 * math.pi
    z = np.clip(rs.standard_normal(n) - 1.0, pc_range[2], np.nextafter(pc_range[5], -np.inf))
    pts = np.stack([r * np.cos(th), r * np.sin(th), z,
                    rs.random_sample(n) * 255.0, np.zeros(n)], 1)
    n_out = int(n * frac_outside)
    if n_out:
        idx = rs.choice(n, n_out, replace=False)
        pts[idx, 0] = pc_range[3] + 1.0 + rs.random_sample(n_out) * 10.0
    return pts.astype(np.float32)


def init_like_state_dict(sd, num_heads=8):
    """Parameters as the reference leaves the sampling layers at initialisation
    (spatial_cross_attention_img.py:293-311, decoder.py:208-226): ``sampling_offsets`` weight 0 and
    bias = the compass-direction grid scaled by the point index, ``attention_weights`` all 0 — the
    operating point of the first training steps (and of bench.py).  Everything else is kept."""
    out = dict(sd)
    for name, a in sd.items():
        leaf = name.split('.')[-1]
        if 'attention_weights' in name:
            out[name] = np.zeros_like(a)
        elif 'sampling_offsets' in name and leaf == 'weight':
            out[name] = np.zeros_like(a)
        elif 'sampling_offsets' in name and leaf == 'bias':
            P = a.size // (num_heads * 2)
            th = np.arange(num_heads, dtype=np.float32) * np.float32(2.0 * math.pi / num_heads)
            g = np.stack([np.cos(th), np.sin(th)], -1).astype(np.float32)
            g = g / np.abs(g).max(-1, keepdims=True)
            g = np.tile(g[:, None, :], (1, P, 1)) * np.arange(1, P + 1, dtype=np.float32)[None, :, None]
            out[name] = g.reshape(-1).astype(np.float32)
    return out


def smooth_like_state_dict(sd, seed=0, num_heads=8):
    """Well-conditioned sampling parameters for gradient parity at full size: the reference's initial compass grid
    SHIFTED off the pixel centres (each offset component + a seeded fraction in [0.2, 0.45]: no sample within 0.02 px of
    a kink of the bilinear interpolant on the BEV grids), small seeded ``sampling_offsets`` weights (offsets move by
    ~0.05 px from query to query, so their gradients are exercised) and small non-zero ``attention_weights`` (a
    non-uniform softmax).  Everything else is kept."""
    out = init_like_state_dict(sd, num_heads)
    for name, a in sd.items():
        leaf = name.split('.')[-1]
        rs = np.random.RandomState((int(seed) * 1000003 + zlib.crc32(('smooth:' + name).encode())) & 0x7FFFFFFF)
        if 'sampling_offsets' in name and leaf == 'bias':
            out[name] = (out[name] + rs.uniform(0.2, 0.45, a.shape)).astype(np.float32)
        elif 'sampling_offsets' in name and leaf == 'weight':
            out[name] = (rs.standard_normal(a.shape) * (0.05 / math.sqrt(a.shape[-1]))).astype(np.float32)
        elif 'attention_weights' in name and leaf == 'weight':
            out[name] = (rs.standard_normal(a.shape) * (0.3 / math.sqrt(a.shape[-1]))).astype(np.float32)
        elif 'attention_weights' in name and leaf == 'bias':
            out[name] = (rs.standard_normal(a.shape) * 0.5).astype(np.float32)
    return out


def smooth_maps(x, k=5):
    """Box filter over the last two axes (window k, mean over the in-range part, times k): feature
    maps with the spatial correlation backbone outputs have, instead of i.i.d. pixels."""
    x = np.asarray(x, np.float64)
    h, w = x.shape[-2:]
    r = k // 2

    def box(a, axis, n):
        c = np.cumsum(np.concatenate([np.zeros_like(np.take(a, [0], axis)), a], axis), axis)
        hi = np.minimum(np.arange(n) + r + 1, n)
        lo = np.maximum(np.arange(n) - r, 0)
        return np.take(c, hi, axis) - np.take(c, lo, axis), (hi - lo)
    s, ch = box(x, x.ndim - 2, h)
    s, cw = box(s, x.ndim - 1, w)
    return (s / (ch[:, None] * cw[None, :]) * k).astype(np.float32)
