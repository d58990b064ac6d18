"""The oracle (oracle/unibev_ref.py) against golden vectors recorded from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from _util import golden, t, tq, metas_from, encoder_case, variant_case, checksum
from oracle import unibev_ref as R
from unibev_amd import synthetic as syn

import make_golden as mg


@pytest.mark.parametrize('case', [c[0] for c in mg.MSDA_CASES])
def test_msda_forward_and_grads(case):
    g = golden('msda')
    v, l, w = (t(g[f'{case}_value']).double().requires_grad_(),
               t(g[f'{case}_loc']).double().requires_grad_(),
               t(g[f'{case}_w']).double().requires_grad_())
    o = R.msda(v, g[f'{case}_shapes'], l, w)
    np.testing.assert_allclose(o.detach().numpy(), g[f'{case}_out64'], rtol=1e-12, atol=1e-12)
    o.backward(t(g[f'{case}_gout']).double())
    np.testing.assert_allclose(v.grad.numpy(), g[f'{case}_gvalue'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(l.grad.numpy(), g[f'{case}_gloc'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(w.grad.numpy(), g[f'{case}_gw'], rtol=1e-10, atol=1e-12)
    o32 = R.msda(t(g[f'{case}_value']), g[f'{case}_shapes'], t(g[f'{case}_loc']), t(g[f'{case}_w']))
    np.testing.assert_allclose(o32.numpy(), g[f'{case}_out'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('case', [c[0] for c in mg.MSDA_CASES])
def test_msda_c_oracle_vs_reference_vectors(case):
    """oracle/msda_ref.c (the C restatement the k1 GPU tests check against) is itself pinned to the vectors recorded
    from the reference's own ``multi_scale_deformable_attn_pytorch`` (msda.npz: f64 output and gradients of the f32
    inputs): f32 inputs, f64 accumulation inside -> agreement to f32 round-off of the outputs."""
    from oracle import c_ref
    g = golden('msda')
    v, l, w, go = g[f'{case}_value'], g[f'{case}_loc'], g[f'{case}_w'], g[f'{case}_gout']
    out = c_ref.msda_forward(v, g[f'{case}_shapes'], l, w)
    np.testing.assert_allclose(out, g[f'{case}_out64'], rtol=2e-6, atol=2e-6)
    gv, gl, gw = c_ref.msda_backward(v, g[f'{case}_shapes'], l, w, go)
    np.testing.assert_allclose(gv, g[f'{case}_gvalue'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gl, g[f'{case}_gloc'], rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(g[f'{case}_gloc']).max())))
    np.testing.assert_allclose(gw, g[f'{case}_gw'], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('tag', ['small', 'mid'])
def test_reference_points_and_point_sampling(tag):
    g = golden('point_sampling')
    H, W, D, bs, nc, ih, iw = [int(x) for x in g[f'{tag}_meta']]
    ref3d = R.get_reference_points(H, W, 8, D, '3d', bs)
    ref2d = R.get_reference_points(H, W, dim='2d', bs=bs)
    np.testing.assert_array_equal(ref3d.numpy(), g[f'{tag}_ref3d'])
    np.testing.assert_array_equal(ref2d.numpy(), g[f'{tag}_ref2d'])
    metas = metas_from(g[f'{tag}_lidar2img'], (ih, iw))
    cam, mask = R.point_sampling_img(ref3d, [-54, -54, -5, 54, 54, 3], metas)
    np.testing.assert_array_equal(mask.numpy(), g[f'{tag}_mask'])
    np.testing.assert_allclose(cam.numpy(), g[f'{tag}_cam'], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(R.point_sampling_pts(ref3d).numpy(), g[f'{tag}_rpl'])


def test_fullsize_visibility_of_synthetic_rig():
    g = golden('point_sampling')
    ref3d = R.get_reference_points(200, 200, 8, 4, '3d', 1)
    cam, mask = R.point_sampling_img(ref3d, [-54, -54, -5, 54, 54, 3],
                                     syn.img_metas(1, 6, (256, 704)))
    vis = mask[:, 0].sum(-1) > 0
    np.testing.assert_array_equal(vis.sum(-1).numpy(), g['full_visible_per_cam'])
    np.testing.assert_array_equal(np.packbits(mask.numpy().reshape(-1)), g['full_mask_bits'])


def test_sca_modules():
    g = golden('sca')
    C, nc, bs, H, W, D, fh, fw, ih, iw = [int(x) for x in g['img_meta']]
    Nq = H * W
    named = [('output_proj.weight', (C, C)), ('output_proj.bias', (C,)),
             ('deformable_attention.sampling_offsets.weight', (128, C)),
             ('deformable_attention.sampling_offsets.bias', (128,)),
             ('deformable_attention.attention_weights.weight', (64, C)),
             ('deformable_attention.attention_weights.bias', (64,)),
             ('deformable_attention.value_proj.weight', (C, C)),
             ('deformable_attention.value_proj.bias', (C,))]
    P = R.state_dict_to_torch(syn.seeded_state_dict(named, 11))
    query = syn.seeded_array('sca_img:query', (bs, Nq, C), 11)
    value = syn.seeded_array('sca_img:value', (nc, fh * fw, bs, C), 11)
    np.testing.assert_array_equal(checksum(query), g['img_query_ck'])
    o = R.sca_img(P, '', t(query), t(value), t(g['img_cam']), t(g['img_mask']), [(fh, fw)], nc)
    np.testing.assert_allclose(o.numpy(), g['img_out'], rtol=1e-5, atol=1e-5)

    C, bs, H, W, D, fh, fw = [int(x) for x in g['pts_meta']]
    P = R.state_dict_to_torch(syn.seeded_state_dict(named, 12))
    query = syn.seeded_array('sca_pts:query', (bs, Nq, C), 12)
    value = syn.seeded_array('sca_pts:value', (fh * fw, bs, C), 12)
    ref3d = R.get_reference_points(H, W, 8, D, '3d', bs)
    o = R.sca_pts(P, '', t(query), t(value), R.point_sampling_pts(ref3d), [(fh, fw)])
    np.testing.assert_allclose(o.numpy(), g['pts_out'], rtol=1e-5, atol=1e-5)


def run_oracle(cfg, sd, inp):
    P = R.state_dict_to_torch(sd)
    return R.transformer_encode_fuse(
        P, cfg, None if inp['img'] is None else [t(x) for x in inp['img']],
        None if inp['pts'] is None else [t(x) for x in inp['pts']],
        tq(inp['bev_q']), inp['bev_h'], inp['bev_w'], t(inp['bev_pos']), inp['metas'],
        return_parts=True)


@pytest.mark.parametrize('name', list(mg.ENCODER_CASES))
def test_encoder_and_fusion(name):
    cfg, sd, inp, g = encoder_case(name)
    fused, (img_bev, pts_bev) = run_oracle(cfg, sd, inp)
    if img_bev is not None:
        np.testing.assert_allclose(img_bev.numpy(), g['img_bev'], rtol=2e-5, atol=2e-5)
    if pts_bev is not None:
        np.testing.assert_allclose(pts_bev.numpy(), g['pts_bev'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(fused.numpy(), g['fused'], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('name', list(mg.VARIANT_CASES))
def test_fusion_variants(name):
    """The feature_norm / use_modal_embeds variants no shipped config selects, under the three modality-flag
    states, against the reference's own outputs (tests/golden/make_golden.py::gen_variants)."""
    cfg, sd, inp, g = variant_case(name)
    P = R.state_dict_to_torch(sd)
    for c_flag, l_flag in mg.VARIANT_FLAGS:
        fused = R.transformer_encode_fuse(P, cfg, [t(x) for x in inp['img']], [t(x) for x in inp['pts']],
                                          t(inp['bev_q']), inp['bev_h'], inp['bev_w'], t(inp['bev_pos']),
                                          inp['metas'], c_flag=c_flag, l_flag=l_flag)
        np.testing.assert_allclose(fused.numpy(), g[f'fused_{c_flag}{l_flag}'], rtol=2e-5, atol=2e-5)


@pytest.mark.slow
def test_encoder_fullsize_statistics():
    cfg, sd, inp, g = encoder_case('fullsize')
    torch.set_num_threads(8)
    fused, (img_bev, pts_bev) = run_oracle(cfg, sd, inp)
    f = fused.numpy().reshape(-1)
    np.testing.assert_allclose(f[g['fused_idx']], g['fused_sub'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(checksum(f)[1], g['fused_ck'][1], rtol=1e-5)
    np.testing.assert_allclose(checksum(img_bev.numpy())[1], g['img_bev_ck'][1], rtol=1e-5)
    np.testing.assert_allclose(checksum(pts_bev.numpy())[1], g['pts_bev_ck'][1], rtol=1e-5)


def test_msda_init_bias_grid():
    g = golden('init')
    np.testing.assert_allclose(R.msda_init_bias(8, 1, 4).numpy(), g['self_bias'], atol=1e-7)
    np.testing.assert_allclose(R.msda_init_bias(8, 1, 8).numpy(), g['cross_bias'], atol=1e-7)
