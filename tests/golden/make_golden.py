#!/usr/bin/env python3
"""Record golden vectors by running the REFERENCE's own modules on CPU.

Container-only: needs ``/root/reference`` (which never travels to the GPU box).  The reference's
hot-path modules are imported unmodified from where they lie, under stand-ins for the un-vendored
mmcv/mmdet symbols (``_ext_stub.py``), fed seeded inputs/parameters (``unibev_amd/synthetic.py``)
and their outputs are written as small ``.npz`` fixtures next to this script.

    python tests/golden/make_golden.py            # regenerate every fixture

The fixtures hold data only: explicit small inputs, seeds for the large ones (with checksums of
the regenerated arrays), and the reference's outputs.
"""
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_MODULES = '/root/reference/projects/UniBEV/unibev_plugin/models/modules'
REF_HEADS = '/root/reference/projects/UniBEV/unibev_plugin/models/dense_heads'
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import _ext_stub as stub                      # noqa: E402
import cfgs                                   # noqa: E402
from unibev_amd import synthetic as syn       # noqa: E402


def load_reference():
    stub.install()
    pkg = types.ModuleType('refmods')
    pkg.__path__ = [REF_MODULES]
    sys.modules['refmods'] = pkg
    mods = {}
    for n in ('spatial_cross_attention_img', 'spatial_cross_attention_pts',
              'encoder_unibev_detr_img', 'encoder_unibev_detr_pts', 'decoder',
              'transformer_fusion'):
        mods[n] = importlib.import_module('refmods.' + n)
    # self-attn slot: the reference's vendored copy of mmcv MultiScaleDeformableAttention
    stub.ATTENTION.register_module(name='MultiScaleDeformableAttention')(
        mods['decoder'].CustomMSDeformableAttention)
    # quirk q10: unregistered name used by unibev_nus_C.py:206
    stub.ATTENTION.register_module(name='MSDeformableAttention3DUniQueryImg')(
        mods['spatial_cross_attention_img'].MSDeformableAttention3DImg)
    return mods


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def checksum(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), float(a.size)])


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


# --------------------------------------------------------------------------- k1 cases
MSDA_CASES = [
    # name, B, shapes, H, Dh, P, Nq, loc spread
    ('c0', 2, [(6, 5)], 8, 32, 4, 37, 0.35),
    ('c1', 1, [(8, 22)], 8, 32, 8, 50, 0.60),      # many out-of-range samples
    ('c2', 2, [(7, 9), (4, 5), (2, 3)], 4, 16, 4, 19, 0.40),
    ('c3', 1, [(3, 4)], 2, 8, 2, 5, 0.50),
    ('c4', 1, [(1, 1)], 8, 16, 8, 9, 0.70),        # degenerate 1x1 map
]


def gen_msda():
    out = {}
    for name, B, shapes, H, Dh, P, Nq, spread in MSDA_CASES:
        L = len(shapes)
        S = sum(h * w for h, w in shapes)
        value = syn.seeded_array(f'msda:{name}:value', (B, S, H, Dh), 1)
        loc = 0.5 + syn.seeded_array(f'msda:{name}:loc', (B, Nq, H, L, P, 2), 1, spread)
        logits = syn.seeded_array(f'msda:{name}:w', (B, Nq, H, L * P), 1)
        w = torch.softmax(t(logits), -1).view(B, Nq, H, L, P).numpy()
        gout = syn.seeded_array(f'msda:{name}:gout', (B, Nq, H * Dh), 1)
        v_, l_, w_ = (t(value).double().requires_grad_(), t(loc).double().requires_grad_(),
                      t(w).double().requires_grad_())
        o = stub.multi_scale_deformable_attn_pytorch(v_, shapes, l_, w_)
        o.backward(t(gout).double())
        o32 = stub.multi_scale_deformable_attn_pytorch(t(value), shapes, t(loc), t(w))
        out.update({f'{name}_shapes': np.asarray(shapes, np.int64), f'{name}_value': value,
                    f'{name}_loc': loc.astype(np.float32), f'{name}_w': w.astype(np.float32),
                    f'{name}_gout': gout, f'{name}_out': o32.numpy(),
                    f'{name}_out64': o.detach().numpy(),
                    f'{name}_gvalue': v_.grad.numpy(), f'{name}_gloc': l_.grad.numpy(),
                    f'{name}_gw': w_.grad.numpy()})
    save('msda', **out)


# --------------------------------------------------------------------------- reference points
def gen_point_sampling(mods):
    enc = mods['encoder_unibev_detr_img'].ImgEncoder
    ptsenc = mods['encoder_unibev_detr_pts'].PtsEncoder
    out = {}
    for tag, (H, W, D, bs, nc, hw) in dict(
            small=(6, 5, 4, 2, 3, (64, 96)), mid=(20, 24, 4, 2, 6, (256, 704))).items():
        ref3d = enc.get_reference_points(H, W, 8, D, dim='3d', bs=bs, device='cpu',
                                         dtype=torch.float32)
        ref2d = enc.get_reference_points(H, W, dim='2d', bs=bs, device='cpu', dtype=torch.float32)
        metas = syn.img_metas(bs, nc, hw, jitter_seed=3)
        self_ = types.SimpleNamespace()
        cam, mask = enc.point_sampling(self_, ref3d, cfgs.PC_RANGE, metas)
        rpl, m2 = ptsenc.point_sampling(self_, ref3d)
        out.update({f'{tag}_meta': np.array([H, W, D, bs, nc, hw[0], hw[1]]),
                    f'{tag}_lidar2img': np.asarray([m['lidar2img'] for m in metas]),
                    f'{tag}_ref3d': ref3d.numpy(), f'{tag}_ref2d': ref2d.numpy(),
                    f'{tag}_cam': cam.numpy(), f'{tag}_mask': mask.numpy(),
                    f'{tag}_rpl': rpl.contiguous().numpy()})
    # full size: per-camera visibility statistics of the synthetic rig (bs=1, 200x200, 6 cams)
    ref3d = enc.get_reference_points(200, 200, 8, 4, dim='3d', bs=1, device='cpu',
                                     dtype=torch.float32)
    metas = syn.img_metas(1, 6, (256, 704))
    cam, mask = enc.point_sampling(types.SimpleNamespace(), ref3d, cfgs.PC_RANGE, metas)
    vis = mask[:, 0].sum(-1) > 0
    out['full_visible_per_cam'] = vis.sum(-1).numpy()
    out['full_visible_any'] = np.array([(vis.sum(0) > 0).sum().item(), (vis.sum(0) > 1).sum().item()])
    out['full_mask_bits'] = np.packbits(mask.numpy().reshape(-1))
    out['full_cam_checksum'] = checksum(cam.numpy()[mask.numpy()])
    save('point_sampling', **out)


# --------------------------------------------------------------------------- module-level cases
def seeded_load(module, seed):
    named = [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
    sd = syn.seeded_state_dict(named, seed)
    module.load_state_dict({k: t(v) for k, v in sd.items()})
    return named


def gen_sca(mods):
    out = {}
    C, nc, bs, H, W, D = 128, 3, 2, 9, 7, 4
    fh, fw = 5, 6
    Nq = H * W
    enc = mods['encoder_unibev_detr_img'].ImgEncoder
    ref3d = enc.get_reference_points(H, W, 8, D, dim='3d', bs=bs, device='cpu', dtype=torch.float32)
    metas = syn.img_metas(bs, nc, (64, 96), jitter_seed=5)
    cam, mask = enc.point_sampling(types.SimpleNamespace(), ref3d, cfgs.PC_RANGE, metas)
    sca = stub.build_attention(dict(
        type='SpatialCrossAttentionImg', pc_range=cfgs.PC_RANGE, num_cams=nc, embed_dims=C,
        batch_first=True,
        deformable_attention=dict(type='MSDeformableAttention3DImg', embed_dims=C, num_points=8,
                                  num_levels=1))).eval()
    seeded_load(sca, 11)
    query = syn.seeded_array('sca_img:query', (bs, Nq, C), 11)
    value = syn.seeded_array('sca_img:value', (nc, fh * fw, bs, C), 11)
    ss = torch.tensor([[fh, fw]])
    with torch.no_grad():
        o = sca(t(query), t(value), t(value), reference_points_cam=cam, bev_mask=mask,
                spatial_shapes=ss, level_start_index=torch.tensor([0]))
    out.update(img_meta=np.array([C, nc, bs, H, W, D, fh, fw, 64, 96]),
               img_lidar2img=np.asarray([m['lidar2img'] for m in metas]),
               img_cam=cam.numpy(), img_mask=mask.numpy(), img_out=o.numpy(),
               img_query_ck=checksum(query), img_value_ck=checksum(value))
    # pts
    fh, fw = 8, 6
    ptsenc = mods['encoder_unibev_detr_pts'].PtsEncoder
    rpl, _ = ptsenc.point_sampling(types.SimpleNamespace(), ref3d)
    scap = stub.build_attention(dict(
        type='SpatialCrossAttentionPts', pc_range=cfgs.PC_RANGE, embed_dims=C, batch_first=True,
        deformable_attention=dict(type='MSDeformableAttention3DPts', embed_dims=C, num_points=8,
                                  num_levels=1))).eval()
    seeded_load(scap, 12)
    query = syn.seeded_array('sca_pts:query', (bs, Nq, C), 12)
    value = syn.seeded_array('sca_pts:value', (fh * fw, bs, C), 12)
    with torch.no_grad():
        o = scap(t(query), t(value), t(value), reference_points_lidar=rpl,
                 spatial_shapes=torch.tensor([[fh, fw]]), level_start_index=torch.tensor([0]))
    out.update(pts_meta=np.array([C, bs, H, W, D, fh, fw]), pts_out=o.numpy(),
               pts_query_ck=checksum(query), pts_value_ck=checksum(value))
    save('sca', **out)


# --------------------------------------------------------------------------- whole encoder+fusion
ENCODER_CASES = {
    # name: (cfg kwargs, bev_h, bev_w, bs, img feat hw, pts feat hw, img_hw, seed)
    'cnw': (dict(embed_dims=128, num_layers=2, num_cams=2), 10, 12, 2, (4, 6), (9, 11), (64, 96), 21),
    'avg': (dict(embed_dims=128, num_layers=1, num_cams=2, fusion_method='avg',
                 feature_norm=None), 10, 12, 2, (4, 6), (9, 11), (64, 96), 22),
    'cat': (dict(embed_dims=128, num_layers=1, num_cams=2, fusion_method='cat',
                 feature_norm=None), 10, 12, 2, (4, 6), (9, 11), (64, 96), 23),
    'cnw256': (dict(embed_dims=256, num_layers=1, num_cams=3), 8, 9, 1, (3, 5), (7, 8), (64, 96), 24),
    'C': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm=None, drop_modality=None,
               modalities='C', img_da_type='MSDeformableAttention3DUniQueryImg'),
          10, 12, 2, (4, 6), None, (64, 96), 25),
    'L': (dict(embed_dims=128, num_layers=1, feature_norm=None, drop_modality=None,
               modalities='L'), 10, 12, 2, None, (9, 11), (64, 96), 26),
    'spatial': (dict(embed_dims=128, num_layers=1, num_cams=2, spatial_norm='SpatialNormWeights',
                     bev_h=10, bev_w=12), 10, 12, 2, (4, 6), (9, 11), (64, 96), 27),
    # two BEV query tables, one per modality (transformer_fusion.py:493-496; shipped config
    # unibev_nus_LC_cnw_dual_queries_modality_dropout.py): `bev_q` is the list [img table, pts table]
    'dual': (dict(embed_dims=128, num_layers=2, num_cams=2, dual_queries=True), 10, 12, 2, (4, 6), (9, 11),
             (64, 96), 28),
}


def encoder_inputs(name, kw, bev_h, bev_w, bs, img_hw_f, pts_hw_f, img_hw, seed):
    C = kw['embed_dims']
    s = 2 if kw.get('fusion_method') == 'cat' else 1
    nc = kw.get('num_cams', 6)
    img = None if img_hw_f is None else [syn.seeded_array(f'enc:{name}:img', (bs, nc, C) + img_hw_f, seed)]
    pts = None if pts_hw_f is None else [syn.seeded_array(f'enc:{name}:pts', (bs, C) + pts_hw_f, seed)]
    bev_q = syn.seeded_array(f'enc:{name}:bev_q', (bev_h * bev_w, C), seed)
    if kw.get('dual_queries'):
        bev_q = [bev_q, syn.seeded_array(f'enc:{name}:bev_q_pts', (bev_h * bev_w, C), seed)]
    bev_pos = syn.seeded_array(f'enc:{name}:bev_pos', (bs, C, bev_h, bev_w), seed)
    oq = syn.seeded_array(f'enc:{name}:oq', (7, 2 * C * s), seed)
    metas = syn.img_metas(bs, nc, img_hw, jitter_seed=seed)
    return img, pts, bev_q, bev_pos, oq, metas


def run_reference_transformer(mods, name, case):
    kw, bev_h, bev_w, bs, img_hw_f, pts_hw_f, img_hw, seed = case
    cfg = cfgs.transformer_cfg(**kw)
    T = mods['transformer_fusion'].UniBEVTransformer
    args = dict(cfg)
    args.pop('type')
    model = T(**args)
    model.init_weights()
    named = seeded_load(model, seed)
    model.eval()
    img, pts, bev_q, bev_pos, oq, metas = encoder_inputs(name, *case)
    parts = {}
    if hasattr(model, 'img_bev_encoder'):
        model.img_bev_encoder.register_forward_hook(lambda m, i, o: parts.__setitem__('img', o))
    if hasattr(model, 'pts_bev_encoder'):
        model.pts_bev_encoder.register_forward_hook(lambda m, i, o: parts.__setitem__('pts', o))
    with torch.no_grad():
        fused, _, _, _ = model(
            None if img is None else [t(x) for x in img],
            None if pts is None else [t(x) for x in pts],
            [t(q) for q in bev_q] if isinstance(bev_q, list) else t(bev_q), t(oq), bev_h, bev_w,
            bev_pos=t(bev_pos), img_metas=metas)
    return cfg, named, fused, parts, (img, pts, bev_q, bev_pos, metas)


def gen_encoders(mods, only=None):
    for name, case in ENCODER_CASES.items():
        if only is not None and name not in only:
            continue
        cfg, named, fused, parts, (img, pts, bev_q, bev_pos, metas) = \
            run_reference_transformer(mods, name, case)
        arrays = dict(cfg_json=np.array(json.dumps(cfg)),
                      param_names=np.array([n for n, _ in named]),
                      param_shapes=np.array([json.dumps(list(s)) for _, s in named]),
                      lidar2img=np.asarray([m['lidar2img'] for m in metas]),
                      fused=fused.numpy(), bev_q_ck=checksum(bev_q), bev_pos_ck=checksum(bev_pos))
        if img is not None:
            arrays['img_ck'] = checksum(img[0])
            arrays['img_bev'] = parts['img'].numpy()
        if pts is not None:
            arrays['pts_ck'] = checksum(pts[0])
            arrays['pts_bev'] = parts['pts'].numpy()
        save('encoder_' + name, **arrays)


FULLSIZE_CASES = {
    # name: (case, profile).  profile 'random': i.i.d. maps + seeded random sampling weights (offsets of
    # a few pixels driven by the queries: every rounding upstream moves sampling points — the
    # adversarial case); 'init': the reference's initial sampling parameters + box-filtered maps.
    'fullsize': ((dict(embed_dims=256, num_layers=3), 200, 200, 1, (8, 22), (180, 180), (256, 704), 31), 'random'),
    'fullsize_init': ((dict(embed_dims=256, num_layers=3), 200, 200, 1, (8, 22), (180, 180), (256, 704), 33), 'init'),
    # gradient-parity fixture: box-filtered maps (d/d(location) nearly continuous across the bilinear kinks) and
    # sampling parameters shifted off the pixel centres (synthetic.smooth_like_state_dict): well conditioned
    'fullsize_smooth': ((dict(embed_dims=256, num_layers=3), 200, 200, 1, (8, 22), (180, 180), (256, 704), 34), 'smooth'),
    # cfg5: cat fusion, C = 128, 800x1440 images -> 25x45 maps
    'fullsize_cat128': ((dict(embed_dims=128, num_layers=3, fusion_method='cat', feature_norm=None),
                         200, 200, 1, (25, 45), (180, 180), (800, 1440), 32), 'random'),
}


def fullsize_inputs(name):
    """(img, pts, bev_q, bev_pos, oq, metas) of a full-size case, profile applied."""
    case, profile = FULLSIZE_CASES[name]
    img, pts, bev_q, bev_pos, oq, metas = encoder_inputs('full' if name == 'fullsize' else name, *case)
    if profile in ('init', 'smooth'):
        img = [syn.smooth_maps(x) for x in img]
        pts = [syn.smooth_maps(x) for x in pts]
    return img, pts, bev_q, bev_pos, oq, metas


def fullsize_state_dict(name, named):
    case, profile = FULLSIZE_CASES[name]
    sd = syn.seeded_state_dict(named, case[-1])
    if profile == 'smooth':
        return syn.smooth_like_state_dict(sd, case[-1])
    return syn.init_like_state_dict(sd) if profile == 'init' else sd


def gen_fullsize(mods, only=None):
    """BASELINE shapes at bs=1 (200x200 BEV, 3 layers): statistics + a strided subsample of the
    reference's output."""
    T = mods['transformer_fusion'].UniBEVTransformer
    for name, (case, profile) in FULLSIZE_CASES.items():
        if only is not None and name not in only:
            continue
        kw, bev_h, bev_w, bs = case[:4]
        cfg = cfgs.transformer_cfg(**kw)
        args = dict(cfg)
        args.pop('type')
        model = T(**args)
        model.init_weights()
        named = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict({k: t(v) for k, v in fullsize_state_dict(name, named).items()})
        model.eval()
        img, pts, bev_q, bev_pos, oq, metas = fullsize_inputs(name)
        parts = {}
        model.img_bev_encoder.register_forward_hook(lambda m, i, o: parts.__setitem__('img', o))
        model.pts_bev_encoder.register_forward_hook(lambda m, i, o: parts.__setitem__('pts', o))
        with torch.no_grad():
            fused, _, _, _ = model([t(x) for x in img], [t(x) for x in pts], t(bev_q), t(oq), bev_h, bev_w,
                                   bev_pos=t(bev_pos), img_metas=metas)
        f = fused.numpy().reshape(-1)
        idx = np.arange(0, f.size, 2503)
        save('encoder_' + name, cfg_json=np.array(json.dumps(cfg)),
             param_names=np.array([n for n, _ in named]),
             param_shapes=np.array([json.dumps(list(s)) for _, s in named]),
             fused_idx=idx, fused_sub=f[idx], fused_ck=checksum(f),
             fused_std=np.array([f.std()]),
             img_bev_ck=checksum(parts['img'].numpy()), pts_bev_ck=checksum(parts['pts'].numpy()),
             img_ck=checksum(img[0]), pts_ck=checksum(pts[0]))


def gen_modality_dropout(mods):
    """Train-mode modality-dropout flags (transformer_fusion.py:463-477, np.random driven)."""
    T = mods['transformer_fusion'].UniBEVTransformer
    out = {}
    for tag, dm in dict(float=0.5, dict=dict(dropout_prob=0.6, lidar_prob=0.3)).items():
        cfg = cfgs.transformer_cfg(embed_dims=128, num_layers=1, num_cams=2, drop_modality=dm)
        args = dict(cfg)
        args.pop('type')
        model = T(**args)
        model.init_weights()
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        case = ENCODER_CASES['cnw']
        img, pts, bev_q, bev_pos, oq, metas = encoder_inputs('cnw', *case)
        np.random.seed(1234)
        flags = []
        with torch.no_grad():
            for _ in range(12):
                model([t(x) for x in img], [t(x) for x in pts], t(bev_q), t(oq), 10, 12,
                      bev_pos=t(bev_pos), img_metas=metas)
                flags.append((int(model.c_flag), int(model.l_flag)))
        out[tag + '_flags'] = np.asarray(flags)
    save('modality_dropout', **out)


# --------------------------------------------------------------------------- fusion variants no shipped config selects
VARIANT_CASES = {
    # name: (cfg kwargs, bev_h, bev_w, bs, img feat hw, pts feat hw, img_hw, seed)
    # transformer_fusion.py:136-155 (learned per-sample channel weights), :152-155 + 287-300 (modality
    # projection under cat fusion), :172-180 + 304-310 (modal embeddings)
    'mlp_cnw': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm='MLP_ChannelNormWeights',
                     bev_h=6, bev_w=8), 6, 8, 2, (4, 6), (9, 11), (64, 96), 41),
    'leaky_cnw': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm='Leaky_ReLU_MLP_ChannelNormWeights',
                       bev_h=6, bev_w=8), 6, 8, 2, (4, 6), (9, 11), (64, 96), 42),
    'elu_cnw': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm='ELU_MLP_ChannelNormWeights',
                     bev_h=6, bev_w=8), 6, 8, 2, (4, 6), (9, 11), (64, 96), 43),
    'sigmoid_cnw': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm='Sigmoid_MLP_ChannelNormWeights',
                         fusion_method='avg', bev_h=6, bev_w=8), 6, 8, 2, (4, 6), (9, 11), (64, 96), 44),
    'modproj': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm='ModalityProjection',
                     fusion_method='cat', bev_h=6, bev_w=8), 6, 8, 2, (4, 6), (9, 11), (64, 96), 45),
    'modproj_spatial': (dict(embed_dims=128, num_layers=1, num_cams=2, feature_norm='ModalityProjection',
                             fusion_method='cat', spatial_norm='SpatialNormWeights', bev_h=6, bev_w=8),
                        6, 8, 2, (4, 6), (9, 11), (64, 96), 46),
    'modal_mlp': (dict(embed_dims=128, num_layers=1, num_cams=2, use_modal_embeds='MLP', bev_h=6, bev_w=8),
                  6, 8, 2, (4, 6), (9, 11), (64, 96), 47),
    'modal_fixed': (dict(embed_dims=128, num_layers=1, num_cams=2, use_modal_embeds='Fixed', feature_norm=None,
                         fusion_method='avg', bev_h=6, bev_w=8), 6, 8, 2, (4, 6), (9, 11), (64, 96), 48),
}
VARIANT_FLAGS = ((1, 1), (1, 0), (0, 1))


def gen_variants(mods):
    """fused_bev_embed of the experimental feature_norm / use_modal_embeds variants under the three
    modality-flag states.  (1, 1) is the eval forward; the single-modality states are train-mode forwards
    (every Dropout at p = 0) whose two ``get_probability`` draws are scripted — the flags are
    otherwise only reachable through np.random (transformer_fusion.py:463-477)."""
    T = mods['transformer_fusion'].UniBEVTransformer
    if not torch.cuda.is_available():
        # the reference builds its flag vectors with ``.cuda()`` (:296-297, 305): identity on this CPU-only box
        torch.Tensor.cuda = lambda self, *a, **k: self
    for name, case in VARIANT_CASES.items():
        kw, bev_h, bev_w, bs, img_hw_f, pts_hw_f, img_hw, seed = case
        cfg = cfgs.transformer_cfg(**kw)
        args = dict(cfg)
        args.pop('type')
        model = T(**args)
        model.init_weights()
        named = seeded_load(model, seed)
        img, pts, bev_q, bev_pos, oq, metas = encoder_inputs(name, *case)
        arrays = dict(cfg_json=np.array(json.dumps(cfg)),
                      param_names=np.array([n for n, _ in named]),
                      param_shapes=np.array([json.dumps(list(s)) for _, s in named]),
                      img_ck=checksum(img[0]), pts_ck=checksum(pts[0]), bev_q_ck=checksum(bev_q))
        for c_flag, l_flag in VARIANT_FLAGS:
            if (c_flag, l_flag) == (1, 1):
                model.eval()
            else:
                model.train()
                for m in model.modules():
                    if isinstance(m, torch.nn.Dropout):
                        m.p = 0.0
                model.drop_modality = 0.5
                script = [True, bool(l_flag)]
                model.get_probability = lambda prob, s=script: s.pop(0)
            with torch.no_grad():
                fused, _, _, _ = model([t(x) for x in img], [t(x) for x in pts], t(bev_q), t(oq), bev_h, bev_w,
                                       bev_pos=t(bev_pos), img_metas=metas)
            assert (int(model.c_flag), int(model.l_flag)) == (c_flag, l_flag)
            arrays[f'fused_{c_flag}{l_flag}'] = fused.numpy().copy()
        save('variant_' + name, **arrays)


def gen_init(mods):
    """init_weights() facts that do not depend on torch's RNG: the sampling_offsets bias grid."""
    T = mods['transformer_fusion'].UniBEVTransformer
    cfg = cfgs.transformer_cfg(embed_dims=128, num_layers=1, num_cams=2)
    args = dict(cfg)
    args.pop('type')
    model = T(**args)
    model.init_weights()
    sd = model.state_dict()
    pre = 'img_bev_encoder.layers.0.attentions.'
    save('init', self_bias=sd[pre + '0.sampling_offsets.bias'].numpy(),
         cross_bias=sd[pre + '1.deformable_attention.sampling_offsets.bias'].numpy(),
         self_w_abs=np.array([sd[pre + '0.sampling_offsets.weight'].abs().sum().item(),
                              sd[pre + '0.attention_weights.weight'].abs().sum().item(),
                              sd[pre + '0.attention_weights.bias'].abs().sum().item()]))


# --------------------------------------------------------------------------- decoder + head (f1)
HEAD_CASES = {
    # name: (head kwargs, transformer kwargs, bs, img feat hw, pts feat hw, img_hw, seed)
    'cnw': (dict(embed_dims=128, bev_h=10, bev_w=12, num_query=9, decoder_layers=2),
            dict(num_layers=1, num_cams=2), 2, (4, 6), (9, 11), (64, 96), 41),
    'cat': (dict(embed_dims=128, bev_h=10, bev_w=12, num_query=7, decoder_layers=2, with_box_refine=False),
            dict(num_layers=1, num_cams=2, fusion_method='cat', feature_norm=None), 1, (4, 6), (9, 11),
            (64, 96), 42),
    # bev_embedding_img / bev_embedding_pts (unibev_head.py:126-133, 172-174)
    'dual': (dict(embed_dims=128, bev_h=10, bev_w=12, num_query=8, decoder_layers=1),
             dict(num_layers=1, num_cams=2, dual_queries=True), 2, (4, 6), (9, 11), (64, 96), 43),
}


def head_inputs(name, hkw, tkw, bs, img_hw_f, pts_hw_f, img_hw, seed):
    C, nc = hkw['embed_dims'], tkw.get('num_cams', 6)
    img = [syn.seeded_array(f'head:{name}:img', (bs, nc, C) + img_hw_f, seed)]
    pts = [syn.seeded_array(f'head:{name}:pts', (bs, C) + pts_hw_f, seed)]
    return img, pts, syn.img_metas(bs, nc, img_hw, jitter_seed=seed)


def gen_head(mods, only=None):
    """UniBEV_Head.forward (models/dense_heads/unibev_head.py:145-242) with the reference's
    DetectionTransformerDecoder / CustomMSDeformableAttention (models/modules/decoder.py:51-338),
    eval mode, seeded parameters: class scores, box predictions, decoder states and references."""
    pkg = types.ModuleType('refheads')
    pkg.__path__ = [REF_HEADS]
    sys.modules['refheads'] = pkg
    Head = importlib.import_module('refheads.unibev_head').UniBEV_Head
    for name, case in HEAD_CASES.items():
        if only is not None and name not in only:
            continue
        hkw, tkw, bs, img_hw_f, pts_hw_f, img_hw, seed = case
        cfg = cfgs.head_cfg(**hkw, **tkw)
        args = json.loads(json.dumps(cfg))
        args.pop('type')
        head = Head(**args)
        head.init_weights()
        named = seeded_load(head, seed)
        head.eval()
        img, pts, metas = head_inputs(name, *case)
        got = {}
        head.transformer.register_forward_hook(lambda m, i, o: got.__setitem__('t', o))
        with torch.no_grad():
            outs = head([t(x) for x in img], [t(x) for x in pts], metas)
        _, hs, init_ref, inter_ref = got['t']
        save('head_' + name, cfg_json=np.array(json.dumps(cfg)),
             param_names=np.array([n for n, _ in named]),
             param_shapes=np.array([json.dumps(list(s)) for _, s in named]),
             lidar2img=np.asarray([m['lidar2img'] for m in metas]),
             img_ck=checksum(img[0]), pts_ck=checksum(pts[0]),
             bev_embed=outs['bev_embed'].numpy(), all_cls_scores=outs['all_cls_scores'].numpy(),
             all_bbox_preds=outs['all_bbox_preds'].numpy(), hs=hs.numpy(),
             init_reference=init_ref.numpy(), inter_references=inter_ref.numpy())
    if only is not None:
        return
    # init_weights facts that do not depend on torch's RNG (unibev_head.py:137-143)
    head = Head(**{k: v for k, v in json.loads(json.dumps(cfgs.head_cfg(**HEAD_CASES['cnw'][0],
                                                                          **HEAD_CASES['cnw'][1]))).items()
                   if k != 'type'})
    torch.manual_seed(5)
    before = head.positional_encoding.row_embed.weight.detach().clone()
    head.init_weights()
    save('head_init', cls_bias=head.cls_branches[0][-1].bias.detach().numpy(),
         pos_untouched=np.array([float((head.positional_encoding.row_embed.weight == before).all())]))


# --------------------------------------------------------------------------- GridMask (row f4, device part)
GRID_MASK_CASES = [
    # seed, (n, c, h, w), prob
    (0, (6, 3, 32, 88), 1.0),
    (1, (2, 3, 64, 64), 1.0),
    (2, (1, 3, 20, 50), 1.0),
    (3, (6, 3, 256, 704), 0.7),
    (4, (6, 3, 256, 704), 0.7),
    (5, (6, 3, 256, 704), 0.7),
]


GRID_MASK_VARIANTS = [
    # seed, (n, c, h, w), rotate, offset, mode: rotated grids (a PIL nearest-neighbour rotation by randint(rotate)
    # degrees, grid_mask.py:111-114) and the random fill of the masked pixels (:120-122)
    (10, (2, 3, 32, 88), 60, False, 1),
    (11, (2, 3, 64, 64), 360, False, 0),
    (12, (1, 2, 20, 50), 1, True, 1),
    (13, (2, 3, 48, 40), 180, True, 1),
    (14, (1, 3, 37, 53), 360, True, 0),
    (15, (1, 3, 64, 64), 91, False, 1),
]


def gen_grid_mask():
    """models/utils/grid_mask.py GridMask.forward as the detector builds it (unibev_detector.py:75:
    GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)): the mask it multiplies with,
    recorded by passing ones, and the next np.random draw (the RNG protocol)."""
    stub.install()
    spec = importlib.util.spec_from_file_location(
        'ref_grid_mask', os.path.join(REF_MODULES, '..', 'utils', 'grid_mask.py'))
    gm_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gm_mod)
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self              # the reference moves its mask with .cuda()
    out = {}
    try:
        for seed, shape, prob in GRID_MASK_CASES:
            gm = gm_mod.GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=prob).train()
            np.random.seed(seed)
            y = gm(torch.ones(*shape))
            assert (y == y[:1, :1]).all()
            out[f's{seed}_mask'] = y[0, 0].numpy().astype(np.uint8)
            out[f's{seed}_next'] = np.array([np.random.rand()])
            out[f's{seed}_meta'] = np.array(list(shape) + [int(round(prob * 100))])
        for seed, shape, rotate, offset, mode in GRID_MASK_VARIANTS:
            gm = gm_mod.GridMask(True, True, rotate=rotate, offset=offset, ratio=0.5, mode=mode, prob=1.0).train()
            ys = []
            for fill in (1.0, 0.0):                            # y(ones) - y(zeros) = mask, y(zeros) = offset * (1 - mask)
                np.random.seed(seed)
                ys.append(gm(torch.full(shape, fill)))
                nxt = np.random.rand()
            assert (ys[0] == ys[0][:1, :1]).all()
            out[f'v{seed}_mask'] = (ys[0] - ys[1])[0, 0].numpy().astype(np.uint8)
            out[f'v{seed}_fill'] = ys[1][0, 0].numpy()
            out[f'v{seed}_next'] = np.array([nxt])
            out[f'v{seed}_meta'] = np.array(list(shape) + [rotate, int(offset), mode])
            np.random.seed(seed)
            x = torch.from_numpy(syn.seeded_array(f'grid_mask:v{seed}', shape, seed))
            out[f'v{seed}_y_ck'] = checksum(gm(x).numpy())
    finally:
        torch.Tensor.cuda = cuda
    save('grid_mask', **out)


PIPELINE_VIEWS = (3, 30, 50, 7)        # views, h, w, seed: 30x50 pads to 32x64 under size_divisor=32


def pipeline_views():
    n, h, w, seed = PIPELINE_VIEWS
    rs = np.random.RandomState(seed)
    return [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(n)]


def gen_pipelines():
    """datasets/pipelines/transform_3d.py: ``PadMultiViewImage`` (:7-57), ``NormalizeMultiviewImage`` (:60-95) and
    ``CustomCollect3D`` (:199-284) — the reference's own classes, imported from where they lie, run on seeded views
    in the order of the shipped configs (normalize -> pad -> collect, configs/unibev/...cnw...:119-137).  mmcv's three
    image helpers are stand-ins restating the published functions (mmcv/image/geometric.py ``impad`` /
    ``impad_to_multiple``: constant border on the bottom / right; mmcv/image/photometric.py ``imnormalize``: f32 copy,
    BGR->RGB when ``to_rgb``, subtract the mean, multiply by 1 / std computed in f64)."""
    stub.install()
    mmcv = sys.modules['mmcv']

    def impad(img, *, shape=None, padding=None, pad_val=0, padding_mode='constant'):
        assert shape is not None and padding is None and padding_mode == 'constant'
        out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
        out[:img.shape[0], :img.shape[1]] = img
        return out

    def impad_to_multiple(img, divisor, pad_val=0):
        h = int(np.ceil(img.shape[0] / divisor)) * divisor
        w = int(np.ceil(img.shape[1] / divisor)) * divisor
        return impad(img, shape=(h, w), pad_val=pad_val)

    def imnormalize(img, mean, std, to_rgb=True):
        img = img.copy().astype(np.float32)
        mean = np.float64(mean.reshape(1, -1))
        stdinv = 1 / np.float64(std.reshape(1, -1))
        if to_rgb:
            img = np.ascontiguousarray(img[..., ::-1])
        img = img - mean.astype(np.float32)          # cv2.subtract / cv2.multiply work in the array's f32
        return img * stdinv.astype(np.float32)

    mmcv.impad, mmcv.impad_to_multiple, mmcv.imnormalize = impad, impad_to_multiple, imnormalize

    class DataContainer:
        def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
            self.data, self.cpu_only = data, cpu_only
    par = stub._mod('mmcv.parallel', DataContainer=DataContainer)
    mmcv.parallel = par
    pipes = stub.Registry('pipeline')
    for n in ('mmdet.datasets',):
        stub._mod(n).__path__ = []
    stub._mod('mmdet.datasets.builder', PIPELINES=pipes)
    spec = importlib.util.spec_from_file_location(
        'ref_transform_3d', os.path.join(REF_MODULES, '..', '..', 'datasets', 'pipelines', 'transform_3d.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    metas = syn.img_metas(1, PIPELINE_VIEWS[0], (PIPELINE_VIEWS[1], PIPELINE_VIEWS[2]))[0]
    for tag, norm, pad in (('cfg', dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False),
                            dict(size_divisor=32)),
                           ('rgb', dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
                            dict(size=(40, 56), pad_val=2))):
        res = dict(img=pipeline_views(), lidar2img=metas['lidar2img'], sample_idx='tok', unrelated=1,
                   points='PTS', pts_filename='a.bin', box_type_3d='LiDAR')
        res = mod.NormalizeMultiviewImage(**norm)(res)
        res = mod.PadMultiViewImage(**pad)(res)
        data = mod.CustomCollect3D(keys=['points', 'img'])(res)
        assert isinstance(data['img_metas'], DataContainer) and data['img_metas'].cpu_only
        m = data['img_metas'].data
        out[tag + '_img'] = np.stack(data['img'])
        out[tag + '_data_keys'] = np.array(json.dumps(list(data)))
        out[tag + '_meta_keys'] = np.array(json.dumps(list(m)))
        for k in ('img_shape', 'pad_shape', 'ori_shape'):
            out[f'{tag}_{k}'] = np.asarray(m[k])
        out[tag + '_norm_mean'] = m['img_norm_cfg']['mean']
        out[tag + '_norm_std'] = m['img_norm_cfg']['std']
        out[tag + '_norm_to_rgb'] = np.array(int(m['img_norm_cfg']['to_rgb']))
        out[tag + '_pad_fixed_size'] = np.asarray(res['pad_fixed_size'] if res['pad_fixed_size'] is not None else [-1])
        out[tag + '_pad_size_divisor'] = np.asarray(res['pad_size_divisor'] if res['pad_size_divisor'] is not None else -1)
        out[tag + '_repr'] = np.array(json.dumps([repr(mod.PadMultiViewImage(**pad)),
                                                  repr(mod.CustomCollect3D(keys=['img']))]))
    out['views_ck'] = checksum(np.stack(pipeline_views()))
    save('pipelines', **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'pipelines':
        return gen_pipelines()
    if len(sys.argv) > 1 and sys.argv[1] == 'variants':
        torch.manual_seed(0)
        return gen_variants(load_reference())
    if len(sys.argv) > 1 and sys.argv[1] == 'fullsize':            # python make_golden.py fullsize <name> ...
        torch.manual_seed(0)
        torch.set_num_threads(8)
        return gen_fullsize(load_reference(), only=tuple(sys.argv[2:]))
    if len(sys.argv) > 1 and sys.argv[1] == 'dual':
        torch.manual_seed(0)
        mods = load_reference()
        gen_encoders(mods, only=('dual',))
        return gen_head(mods, only=('dual',))
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mods = load_reference()
    gen_msda()
    gen_point_sampling(mods)
    gen_sca(mods)
    gen_encoders(mods)
    gen_modality_dropout(mods)
    gen_variants(mods)
    gen_init(mods)
    gen_fullsize(mods)
    gen_head(mods)
    gen_grid_mask()
    gen_pipelines()


if __name__ == '__main__':
    main()
