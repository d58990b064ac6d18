"""Shared helpers for the parity tests (fixture loading, seeded inputs)."""
import json
import os

import numpy as np
import torch

from unibev_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def t(a, dtype=None, device=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    if device is not None:
        x = x.to(device)
    return x


def tq(q, dtype=None, device=None, grad=False):
    """BEV query table(s) of a fixture: one array, or [img table, pts table] for ``dual_queries``."""
    if isinstance(q, (list, tuple)):
        return [tq(x, dtype, device, grad) for x in q]
    x = t(q, dtype, device)
    return x.requires_grad_() if grad else x


def checksum(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), float(a.size)])


def metas_from(lidar2img, img_hw):
    """Rebuild img_metas from a stored (bs, Nc, 4, 4) lidar2img array."""
    return [dict(lidar2img=[m for m in l2i], img_shape=[(img_hw[0], img_hw[1], 3)] * len(l2i))
            for l2i in np.asarray(lidar2img)]


def encoder_case(name):
    """Regenerate the seeded inputs / parameters of an ``encoder_<name>`` fixture and return
    (cfg, state_dict(np), inputs dict, fixture)."""
    import make_golden as mg          # only for the case table; does not touch /root/reference
    g = golden('encoder_' + name)
    cfg = json.loads(str(g['cfg_json']))
    case = mg.ENCODER_CASES[name] if name in mg.ENCODER_CASES else None
    if name in mg.FULLSIZE_CASES:
        case = mg.FULLSIZE_CASES[name][0]
        kw, bev_h, bev_w, bs, img_hw_f, pts_hw_f, img_hw, seed = case
        img, pts, bev_q, bev_pos, oq, metas = mg.fullsize_inputs(name)
        named = [(n, tuple(json.loads(s))) for n, s in zip(g['param_names'], g['param_shapes'])]
        sd = mg.fullsize_state_dict(name, named)
        np.testing.assert_array_equal(checksum(img[0]), g['img_ck'])
        np.testing.assert_array_equal(checksum(pts[0]), g['pts_ck'])
        inputs = dict(img=img, pts=pts, bev_q=bev_q, bev_pos=bev_pos, metas=metas, bev_h=bev_h,
                      bev_w=bev_w, bs=bs)
        return cfg, sd, inputs, g
    kw, bev_h, bev_w, bs, img_hw_f, pts_hw_f, img_hw, seed = case
    tag = name
    img, pts, bev_q, bev_pos, oq, metas = mg.encoder_inputs(tag, *case)
    named = [(n, tuple(json.loads(s))) for n, s in zip(g['param_names'], g['param_shapes'])]
    sd = syn.seeded_state_dict(named, seed)
    # the regenerated inputs must be the ones the reference saw
    np.testing.assert_array_equal(checksum(bev_q), g['bev_q_ck']) if 'bev_q_ck' in g else None
    if img is not None and 'img_ck' in g:
        np.testing.assert_array_equal(checksum(img[0]), g['img_ck'])
    if pts is not None and 'pts_ck' in g:
        np.testing.assert_array_equal(checksum(pts[0]), g['pts_ck'])
    inputs = dict(img=img, pts=pts, bev_q=bev_q, bev_pos=bev_pos, metas=metas, bev_h=bev_h,
                  bev_w=bev_w, bs=bs)
    return cfg, sd, inputs, g


def variant_case(name):
    """Seeded inputs / parameters of a ``variant_<name>`` fixture (the fusion variants no shipped config
    selects): (cfg, state_dict(np), inputs dict, fixture with ``fused_<c><l>`` per modality-flag state)."""
    import make_golden as mg
    g = golden('variant_' + name)
    cfg = json.loads(str(g['cfg_json']))
    case = mg.VARIANT_CASES[name]
    kw, bev_h, bev_w, bs, img_hw_f, pts_hw_f, img_hw, seed = case
    img, pts, bev_q, bev_pos, oq, metas = mg.encoder_inputs(name, *case)
    named = [(n, tuple(json.loads(s))) for n, s in zip(g['param_names'], g['param_shapes'])]
    sd = syn.seeded_state_dict(named, seed)
    np.testing.assert_array_equal(checksum(img[0]), g['img_ck'])
    np.testing.assert_array_equal(checksum(pts[0]), g['pts_ck'])
    np.testing.assert_array_equal(checksum(bev_q), g['bev_q_ck'])
    inputs = dict(img=img, pts=pts, bev_q=bev_q, bev_pos=bev_pos, metas=metas, bev_h=bev_h, bev_w=bev_w, bs=bs)
    return cfg, sd, inputs, g
