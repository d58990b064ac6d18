"""Module-level parity on the GPU: point sampling, SCA modules and the whole encoder+fusion path
against (i) vectors recorded from the reference and (ii) the oracle on seeded inputs."""
import json

import numpy as np
import pytest
import torch

from _util import golden, t, tq, metas_from, encoder_case, variant_case, checksum
import make_golden as mg
from unibev_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda'
PC = [-54, -54, -5, 54, 54, 3]


@pytest.mark.parametrize('tag', ['small', 'mid'])
def test_point_sampling_vs_reference_vectors(tag):
    from unibev_amd.modules.encoders import ImgEncoder
    g = golden('point_sampling')
    H, W, D, bs, nc, ih, iw = [int(x) for x in g[f'{tag}_meta']]
    ref3d = ImgEncoder.get_reference_points(H, W, 8, D, dim='3d', bs=bs, device=DEV)
    ref2d = ImgEncoder.get_reference_points(H, W, dim='2d', bs=bs, device=DEV)
    np.testing.assert_array_equal(ref3d.cpu().numpy(), g[f'{tag}_ref3d'])
    np.testing.assert_array_equal(ref2d.cpu().numpy(), g[f'{tag}_ref2d'])
    enc = ImgEncoder.__new__(ImgEncoder)
    cam, mask = ImgEncoder.point_sampling(enc, ref3d, PC, metas_from(g[f'{tag}_lidar2img'], (ih, iw)))
    gm = g[f'{tag}_mask']
    # visibility is index-like: must be identical except where a projected coordinate sits within
    # 1e-6 of a boundary (the reference's batched matmul has no defined summation order)
    diff = mask.cpu().numpy() != gm
    assert diff.sum() <= 2, int(diff.sum())
    np.testing.assert_allclose(cam.cpu().numpy()[gm], g[f'{tag}_cam'][gm], rtol=2e-5, atol=2e-6)


def test_point_sampling_full_size_visibility():
    from unibev_amd.modules.encoders import pillar_axes
    from unibev_amd.functional import point_sampling
    g = golden('point_sampling')
    xs, ys, zs = pillar_axes(200, 200, 8, 4, DEV)
    l2i = t(np.asarray([m['lidar2img'] for m in syn.img_metas(1, 6, (256, 704))]), torch.float32, DEV)
    cam, mask, vis0, count = point_sampling(l2i, xs, ys, zs, PC, (256, 704))
    ref_bits = np.unpackbits(g['full_mask_bits'])[:mask.numel()].reshape(mask.shape).astype(bool)
    assert (mask.cpu().numpy() != ref_bits).sum() <= 4
    np.testing.assert_allclose(vis0.sum(-1).cpu().numpy(), g['full_visible_per_cam'], atol=2)
    seen = mask.any(-1)
    torch.testing.assert_close(count, seen.sum(0).clamp(min=1).float())
    assert torch.equal(vis0.bool(), seen[:, 0])


def _load(module, sd):
    missing = module.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    left = [k for k in missing.missing_keys if not k.startswith('decoder')]
    assert not left, left


def _build(cfg):
    from unibev_amd import build_transformer
    cfg = json.loads(json.dumps(cfg))
    return build_transformer(cfg)


def _run(model, inp, dtype=torch.float32):
    img = None if inp['img'] is None else [t(x, dtype, DEV) for x in inp['img']]
    pts = None if inp['pts'] is None else [t(x, dtype, DEV) for x in inp['pts']]
    return model.encode(img, pts, tq(inp['bev_q'], dtype, DEV), inp['bev_h'], inp['bev_w'],
                        bev_pos=t(inp['bev_pos'], dtype, DEV), img_metas=inp['metas'],
                        return_parts=True)


@pytest.mark.parametrize('name', list(mg.ENCODER_CASES))
def test_encoder_fusion_vs_reference_vectors(name):
    """fp32 product path vs fused_bev_embed / per-modality BEV features recorded from the
    reference (same seeded parameters and inputs).  north_star bar 1e-3 rel; fp32 meets 1e-4."""
    cfg, sd, inp, g = encoder_case(name)
    model = _build(cfg).to(DEV).eval()
    _load(model, sd)
    with torch.no_grad():
        fused, img_bev, pts_bev = _run(model, inp)
    scale = np.abs(g['fused']).max()
    if img_bev is not None:
        np.testing.assert_allclose(img_bev.cpu().numpy(), g['img_bev'], rtol=1e-4, atol=1e-4 * scale)
    if pts_bev is not None:
        np.testing.assert_allclose(pts_bev.cpu().numpy(), g['pts_bev'], rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(fused.cpu().numpy(), g['fused'], rtol=1e-4, atol=1e-4 * scale)


@pytest.mark.parametrize('streams', [2, 1])
@pytest.mark.parametrize('name,gemm', [('cnw', 'mfma'), ('cat', 'mfma'), ('cnw', 'library'), ('dual', 'library')])
def test_encoder_gradients_vs_oracle(name, gemm, streams):
    """Backward of the whole path: d(sum(fused * cot)) w.r.t. inputs and every parameter vs torch autograd through
    the oracle — with the two encoders on two HIP streams (the default) and on one.  ``gemm``: 'mfma' = the default
    split-bf16 Linear layers at the 1e-3 bar; 'library' = IEEE f32 GEMMs, where the sampling / normalisation / fusion
    kernels and every gradient route must agree with the oracle to f32 round-off (bars 1e-4 element-wise, 3e-5
    normwise; measured 5e-6).  The two-layer ``dual`` fixture (two query tables) is checked in that mode only: its
    random sampling parameters put points within the split-bf16 round-off of a pixel boundary, where d/d(offset) jumps
    (one flipped point moves every upstream gradient by ~1e-2; seeds 28-31 each flip in one of the two branches, while
    the forward holds 4e-5) — the discontinuity is the reference's own (bilinear kinks), not a routing difference."""
    from oracle import unibev_ref as R
    from unibev_amd.linear import set_f32_gemm
    from unibev_amd.modules import transformer as TR
    was = TR._TWO_STREAMS[0]
    TR.set_two_streams(streams == 2)
    prev = set_f32_gemm(gemm)
    try:
        _encoder_gradients_vs_oracle(name, R, bars=(4e-3, 1e-3) if gemm == 'mfma' else (1e-4, 3e-5))
    finally:
        set_f32_gemm(prev)
        TR.set_two_streams(was)


def _encoder_gradients_vs_oracle(name, R, case=None, flags=(1, 1), bars=(4e-3, 1e-3)):
    cfg, sd, inp, g = case if case is not None else encoder_case(name)
    cot = syn.seeded_array('cot:' + name, g['fused' if case is None else 'fused_11'].shape, 5)
    # oracle
    P = {k: v.requires_grad_() for k, v in R.state_dict_to_torch(sd).items()}
    oi = [t(x).requires_grad_() for x in inp['img']]
    op = [t(x).requires_grad_() for x in inp['pts']]
    oq = tq(inp['bev_q'], grad=True)
    fused_ref = R.transformer_encode_fuse(P, cfg, oi, op, oq, inp['bev_h'], inp['bev_w'],
                                          t(inp['bev_pos']), inp['metas'], c_flag=flags[0], l_flag=flags[1])
    (fused_ref * t(cot)).sum().backward()
    # product
    model = _build(cfg).to(DEV).eval()
    _load(model, sd)
    model.forced_flags = tuple(flags)
    gi = [t(x, device=DEV).requires_grad_() for x in inp['img']]
    gp = [t(x, device=DEV).requires_grad_() for x in inp['pts']]
    gq = tq(inp['bev_q'], device=DEV, grad=True)
    fused = model.encode(gi, gp, gq, inp['bev_h'], inp['bev_w'], bev_pos=t(inp['bev_pos'], device=DEV),
                         img_metas=inp['metas'])
    (fused * t(cot, device=DEV)).sum().backward()

    def close(a, b, what):
        # every Linear of the product path is a split-bf16 MFMA product (~2^-17 per product; since round 3 also the
        # first layer's, whose batch-expanded input used to fall to the IEEE library GEMM): normwise 1e-3 like the
        # bar on the BEV features, and 4e-3 on the single worst element (measured 2.8e-3 on an FFN weight gradient of
        # this random-parameter fixture, which amplifies a unit round-off ~1000x)
        b = b.numpy()
        a = a.cpu().numpy()
        s = max(np.abs(b).max(), 1e-6)
        err = np.abs(a - b).max() / s
        nerr = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
        assert err < bars[0] and nerr < bars[1], (what, err, nerr)
    close(gi[0].grad, oi[0].grad, 'img feats')
    close(gp[0].grad, op[0].grad, 'pts feats')
    if isinstance(gq, list):              # dual_queries: one table per modality
        close(gq[0].grad, oq[0].grad, 'bev queries (img)')
        close(gq[1].grad, oq[1].grad, 'bev queries (pts)')
    else:
        close(gq.grad, oq.grad, 'bev queries')
    for k, p in model.named_parameters():
        if k.startswith('decoder') or k.startswith('reference_points'):
            continue
        if P[k].grad is None:
            # a parameter the flag state switches off (e.g. a modality projection whose half is the real feature):
            # no gradient in the oracle, none or zero here
            assert p.grad is None or not p.grad.any(), k
            continue
        assert p.grad is not None, k
        close(p.grad, P[k].grad, k)


@pytest.mark.parametrize('name', list(mg.VARIANT_CASES))
def test_fusion_variants_vs_reference_vectors(name):
    """The feature_norm / use_modal_embeds variants no shipped config selects (learned per-sample channel weights,
    modality projection, modal embeddings) under the three modality-flag states, against fused_bev_embed recorded
    from the reference (tests/golden/make_golden.py::gen_variants)."""
    cfg, sd, inp, g = variant_case(name)
    model = _build(cfg).to(DEV).eval()
    _load(model, sd)
    for c_flag, l_flag in mg.VARIANT_FLAGS:
        model.forced_flags = (c_flag, l_flag)
        with torch.no_grad():
            fused, _, _ = _run(model, inp)
        ref = g[f'fused_{c_flag}{l_flag}']
        np.testing.assert_allclose(fused.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max(),
                                   err_msg=f'flags {c_flag}{l_flag}')


@pytest.mark.parametrize('name,flags', [('mlp_cnw', (1, 1)), ('elu_cnw', (0, 1)), ('modproj_spatial', (1, 1)),
                                        ('modproj', (1, 0)), ('modal_mlp', (1, 1)), ('modal_fixed', (0, 1))])
def test_fusion_variant_gradients_vs_oracle(name, flags):
    from oracle import unibev_ref as R
    _encoder_gradients_vs_oracle(name, R, case=variant_case(name), flags=flags)


def test_sca_modules_vs_reference_vectors():
    from unibev_amd import build_attention
    from unibev_amd.modules.deform_attn import shapes_tensor
    from unibev_amd.modules.encoders import ImgEncoder, PtsEncoder
    g = golden('sca')
    C, nc, bs, H, W, D, fh, fw, ih, iw = [int(x) for x in g['img_meta']]
    Nq = H * W
    sca = build_attention(dict(type='SpatialCrossAttentionImg', pc_range=PC, num_cams=nc,
                               embed_dims=C, batch_first=True,
                               deformable_attention=dict(type='MSDeformableAttention3DImg',
                                                         embed_dims=C, num_points=8,
                                                         num_levels=1))).to(DEV).eval()
    sd = syn.seeded_state_dict([(k, tuple(v.shape)) for k, v in sca.state_dict().items()], 11)
    sca.load_state_dict({k: t(v) for k, v in sd.items()})
    query = t(syn.seeded_array('sca_img:query', (bs, Nq, C), 11), device=DEV)
    value = t(syn.seeded_array('sca_img:value', (nc, fh * fw, bs, C), 11), device=DEV)
    with torch.no_grad():
        out = sca(query, value, value, reference_points_cam=t(g['img_cam'], device=DEV),
                  bev_mask=t(g['img_mask'], device=DEV), spatial_shapes=shapes_tensor([(fh, fw)], DEV),
                  level_start_index=torch.zeros(1, dtype=torch.long, device=DEV))
        np.testing.assert_allclose(out.cpu().numpy(), g['img_out'], rtol=1e-4, atol=1e-4)
        # the fallback for shapes the fused kernel rejects (k1 per camera over all queries, invisible rows
        # masked) gives the same answer
        slots = sca._masked_path(query, value.permute(2, 0, 1, 3).reshape(bs * nc, fh * fw, C),
                                  t(g['img_cam'], device=DEV), t(g['img_mask'], device=DEV),
                                  shapes_tensor([(fh, fw)], DEV),
                                  torch.zeros(1, dtype=torch.long, device=DEV))
        out2 = sca.output_proj(slots) + query
        np.testing.assert_allclose(out2.cpu().numpy(), g['img_out'], rtol=1e-4, atol=1e-4)

    C, bs, H, W, D, fh, fw = [int(x) for x in g['pts_meta']]
    scap = build_attention(dict(type='SpatialCrossAttentionPts', pc_range=PC, embed_dims=C,
                                batch_first=True,
                                deformable_attention=dict(type='MSDeformableAttention3DPts',
                                                          embed_dims=C, num_points=8,
                                                          num_levels=1))).to(DEV).eval()
    sd = syn.seeded_state_dict([(k, tuple(v.shape)) for k, v in scap.state_dict().items()], 12)
    scap.load_state_dict({k: t(v) for k, v in sd.items()})
    query = t(syn.seeded_array('sca_pts:query', (bs, Nq, C), 12), device=DEV)
    value = t(syn.seeded_array('sca_pts:value', (fh * fw, bs, C), 12), device=DEV)
    ref3d = PtsEncoder.get_reference_points(H, W, 8, D, dim='3d', bs=bs, device=DEV)
    enc = PtsEncoder.__new__(PtsEncoder)
    rpl, _ = PtsEncoder.point_sampling(enc, ref3d)
    with torch.no_grad():
        out = scap(query, value, value, reference_points_lidar=rpl,
                   spatial_shapes=shapes_tensor([(fh, fw)], DEV),
                   level_start_index=torch.zeros(1, dtype=torch.long, device=DEV))
    np.testing.assert_allclose(out.cpu().numpy(), g['pts_out'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('fixture', ['fullsize', 'fullsize_init', 'fullsize_cat128'])
def test_encoder_fullsize_vs_reference_statistics(fixture):
    """BASELINE shapes at bs=1 (200x200 BEV, 3 layers, 180x180 LiDAR feats): cfg4 (6 x 8x22 image
    feats, C=256, CNW) with adversarial and with initial-state sampling parameters, and cfg5 (6 x 25x45
    image feats, C=128, cat): a strided sample and checksums of the reference's fused_bev_embed."""
    cfg, sd, inp, g = encoder_case(fixture)
    model = _build(cfg).to(DEV).eval()
    _load(model, sd)
    with torch.no_grad():
        fused, img_bev, pts_bev = _run(model, inp)
    f = fused.cpu().numpy().reshape(-1)
    scale = np.abs(g['fused_sub']).max()
    np.testing.assert_allclose(f[g['fused_idx']], g['fused_sub'], rtol=1e-3, atol=1e-3 * scale)
    err = np.linalg.norm(f[g['fused_idx']] - g['fused_sub']) / np.linalg.norm(g['fused_sub'])
    assert err < 6e-4, err            # split-bf16 GEMMs: 3.2e-4 measured on the adversarial fixture
    np.testing.assert_allclose(checksum(f)[1], g['fused_ck'][1], rtol=1e-4)
    np.testing.assert_allclose(checksum(img_bev.cpu().numpy())[1], g['img_bev_ck'][1], rtol=1e-4)
    np.testing.assert_allclose(checksum(pts_bev.cpu().numpy())[1], g['pts_bev_ck'][1], rtol=1e-4)


# Full-size distance of the reduced-precision modes to the REFERENCE-recorded vectors (normwise, on the recorded
# subsample) against BASELINE's bar of 1e-3.  Modes: 'fp16' / 'bf16' = autocast (16-bit GEMM operands, value maps and
# residual stream); 'value-fp16' / 'value-bf16' = f32 everywhere (split-bf16 MFMA GEMMs, f32 stream, offsets, logits)
# with ONLY the projected value maps / sampled outputs stored in 16 bits (deform_attn.set_value_storage).
# Measured on MI355X (tools/precision_study.py, profiles/r03_precision_study.txt):
#   fixture            fp32     value-fp16  fp16     value-bf16  bf16
#   fullsize_init      7e-6     2.3e-4      7.2e-4   1.8e-3      5.8e-3   initial sampling parameters, correlated maps
#   fullsize           3.2e-4   3.1e-3      2.8e-2   2.4e-2      1.9e-1   i.i.d. maps, random offset weights
#   fullsize_cat128    6.5e-5   8.6e-4      6.1e-3   6.9e-3      4.7e-2   cfg5 (C = 128, 25x45 maps)
# The adversarial fixture amplifies a unit round-off ~1000x (f32: 6e-8 -> 6e-5): every rounding of a sampled value
# moves the next layers' sampling points, and on i.i.d. maps a moved point changes the sample by O(1).  Only f32
# passes everywhere; value-fp16 passes at the operating point and on cfg5; nothing with bf16 storage passes.
# Combinations that MISS the bar are listed with a ceiling (regression guard) and reported as xfail — strictly: a
# listed combination that comes inside the bar fails the test until it is taken off the list.
PARITY_BAR = 1e-3
LOWP_MODES = {'fp16': (torch.float16, None), 'bf16': (torch.bfloat16, None),
              'value-fp16': (torch.float32, torch.float16), 'value-bf16': (torch.float32, torch.bfloat16)}
LOWP_MISSES = {('fullsize', 'fp16'): 4e-2, ('fullsize_cat128', 'fp16'): 9e-3,
               ('fullsize_init', 'bf16'): 8e-3, ('fullsize', 'bf16'): 0.26, ('fullsize_cat128', 'bf16'): 6.5e-2,
               ('fullsize', 'value-fp16'): 4.5e-3,
               ('fullsize_init', 'value-bf16'): 2.6e-3, ('fullsize', 'value-bf16'): 3.5e-2,
               ('fullsize_cat128', 'value-bf16'): 1e-2}


@pytest.mark.parametrize('mode', list(LOWP_MODES))
@pytest.mark.parametrize('fixture', ['fullsize_init', 'fullsize', 'fullsize_cat128'])
def test_fullsize_reduced_precision_distance_to_reference_vectors(fixture, mode):
    from unibev_amd.modules.deform_attn import set_value_storage
    cfg, sd, inp, g = encoder_case(fixture)
    model = _build(cfg).to(DEV).eval()
    _load(model, sd)
    dtype, vstore = LOWP_MODES[mode]
    prev = set_value_storage(vstore)
    try:
        with torch.no_grad(), torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
            fused, _, _ = _run(model, inp)
    finally:
        set_value_storage(prev)
    assert fused.dtype == dtype
    f = fused.float().cpu().numpy().reshape(-1)[g['fused_idx']]
    err = np.linalg.norm(f - g['fused_sub']) / np.linalg.norm(g['fused_sub'])
    ceiling = LOWP_MISSES.get((fixture, mode))
    if ceiling is None:
        assert err < PARITY_BAR, err
        return
    assert err < ceiling, f'{err:.3e}: worse than the recorded miss'
    assert err >= PARITY_BAR, f'{err:.3e} is now inside the bar: take ({fixture}, {mode}) off LOWP_MISSES'
    pytest.xfail(f'{mode} on {fixture}: {err:.2e} misses the 1e-3 bar (expected, see the table above)')


def test_value_storage_mode_gradients_match_f32():
    """value-fp16 backward: grad_output / grad_value travel in fp16 through the sampling kernels, everything else in
    f32 — parameter and input gradients stay within 3e-2 (normwise; measured 3.3e-3 .. 1.3e-2: unscaled fp16
    gradients of ~1e-5 sit in the type's subnormal range — the mode is a storage study, not the training default) of the all-f32 run on the small CNW case."""
    from unibev_amd.modules.deform_attn import set_value_storage
    cfg, sd, inp, g = encoder_case('cnw')
    grads = {}
    for tag, st in (('f32', None), ('v16', torch.float16)):
        model = _build(cfg).to(DEV).eval()
        _load(model, sd)
        img = [t(x, device=DEV).requires_grad_() for x in inp['img']]
        pts = [t(x, device=DEV).requires_grad_() for x in inp['pts']]
        prev = set_value_storage(st)
        try:
            fused = model.encode(img, pts, t(inp['bev_q'], device=DEV), inp['bev_h'], inp['bev_w'],
                                 bev_pos=t(inp['bev_pos'], device=DEV), img_metas=inp['metas'])
            fused.square().mean().backward()
        finally:
            set_value_storage(prev)
        assert fused.dtype == torch.float32
        grads[tag] = [img[0].grad, pts[0].grad] + [p.grad for n, p in model.named_parameters()
                                                  if p.grad is not None and 'value_proj.weight' in n]
    for a, b in zip(grads['v16'], grads['f32']):
        assert float((a - b).norm() / b.norm()) < 3e-2, float((a - b).norm() / b.norm())


@pytest.mark.parametrize('lowp_stream', [True, False])
@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-2), (torch.bfloat16, 5e-2)])
def test_encoder_autocast_close_to_fp32(dtype, tol, lowp_stream):
    """Mixed precision (autocast: GEMMs and sampled values in 16-bit, LayerNorm statistics /
    softmax / locations in f32; residual stream in 16-bit like the reference's fp16 mode, or f32)
    stays within a stated normwise distance of the fp32 reference output."""
    cfg, sd, inp, g = encoder_case('cnw')
    model = _build(cfg).to(DEV).eval()
    model.lowp_stream = lowp_stream
    _load(model, sd)
    with torch.no_grad(), torch.autocast('cuda', dtype=dtype):
        fused, img_bev, _ = _run(model, inp)
    assert img_bev.dtype == (dtype if lowp_stream else torch.float32)
    ref = g['fused']
    err = np.linalg.norm(fused.float().cpu().numpy() - ref) / np.linalg.norm(ref)
    assert err < tol, err


def test_missing_extension_or_cpu_tensor_fails_loudly():
    from unibev_amd.functional import bev_fuse
    with pytest.raises(RuntimeError):
        bev_fuse(torch.zeros(1, 4, 8), None, torch.ones(8), torch.ones(8))


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_add_dropout_layernorm_16bit_stream(dtype):
    """identity / y / their gradients in the branch's own 16-bit type (stream_dtype == dtype):
    equals the f32-stream kernel fed the same 16-bit values, rounded once at the end."""
    from unibev_amd.functional import add_dropout_layernorm
    torch.manual_seed(1)
    R, C = 777, 256
    x = torch.randn(R, C, device=DEV).to(dtype)
    idn = torch.randn(R, C, device=DEV).to(dtype)
    g = torch.randn(C, device=DEV) * 0.2 + 1
    b = torch.randn(C, device=DEV) * 0.1
    cot = torch.randn(R, C, device=DEV).to(dtype)
    outs = []
    for stream16 in (True, False):
        xa, ga, ba = (v.clone().requires_grad_() for v in (x, g, b))
        ia = (idn if stream16 else idn.float()).clone().requires_grad_()
        y = add_dropout_layernorm(xa, ia, ga, ba, 0.0, training=False)
        assert y.dtype == (dtype if stream16 else torch.float32)
        y.backward(cot if stream16 else cot.float())
        assert ia.grad.dtype == ia.dtype
        outs.append((y, xa.grad, ia.grad, ga.grad, ba.grad))
    (y16, gx16, gi16, gg16, gb16), (y32, gx32, gi32, gg32, gb32) = outs
    # same arithmetic in both instantiations up to fused-multiply-add placement: equal to one
    # unit in the last place of the 16-bit type, and bit-equal almost everywhere
    for a, b in ((y16, y32.to(dtype)), (gx16, gx32), (gi16, gi32.to(dtype))):
        torch.testing.assert_close(a.float(), b.float(), rtol=2.0 ** -7, atol=1e-5)
        assert (a != b).float().mean().item() < 1e-2
    torch.testing.assert_close(gg16, gg32, rtol=1e-5, atol=1e-4)   # atomics: order varies
    torch.testing.assert_close(gb16, gb32, rtol=1e-5, atol=1e-4)


def test_add_dropout_layernorm_backward_column_sums_are_ordered():
    """gamma / beta gradients (and the bias column sums that ride along) of the fused residual + dropout + LayerNorm
    backward at the step's size (80 000 x 256): the blocks' partial sums are added in block order
    (ubv_add_dropout_layernorm_backward ordered_workspace), so repeated passes agree BIT FOR BIT — with one f32 atomic
    per column per block they moved in the last bits from run to run — and they agree with an f64 reduction."""
    from unibev_amd.functional import add_dropout_layernorm
    torch.manual_seed(3)
    R, C = 80000, 256
    x = torch.randn(R, C, device=DEV)
    idn = torch.randn(R, C, device=DEV)
    go = torch.randn(R, C, device=DEV)
    runs = []
    for _ in range(3):
        xa = x.clone().requires_grad_()
        g = (1.0 + 0.1 * torch.randn(C, device=DEV, generator=torch.Generator(DEV).manual_seed(5))).requires_grad_()
        b = torch.zeros(C, device=DEV, requires_grad=True)
        y = add_dropout_layernorm(xa, idn, g, b, 0.0, training=False)
        y.backward(go)
        runs.append((g.grad.clone(), b.grad.clone(), xa.grad.sum(0)))
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1])
    ref_b = go.double().sum(0)
    torch.testing.assert_close(runs[0][1].double(), ref_b, rtol=1e-5, atol=1e-5 * float(ref_b.abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_add_dropout_layernorm_vs_torch(dtype):
    """Fused residual + dropout + LayerNorm: eval mode equals F.layer_norm(x + identity) forward and
    backward; train mode drops ~p of the elements, scales the rest, and its backward is consistent
    with the regenerated mask."""
    from unibev_amd.functional import add_dropout_layernorm
    torch.manual_seed(0)
    R, C = 1000, 256
    x = torch.randn(R, C, device=DEV).to(dtype)
    idn = torch.randn(R, C, device=DEV)
    g = torch.randn(C, device=DEV) * 0.2 + 1
    b = torch.randn(C, device=DEV) * 0.1
    cot = torch.randn(R, C, device=DEV)
    xa, ia, ga, ba = (v.clone().requires_grad_() for v in (x, idn, g, b))
    y = add_dropout_layernorm(xa, ia, ga, ba, 0.1, training=False)
    (y * cot).sum().backward()
    xr, ir, gr, br = (v.clone().float().requires_grad_() for v in (x, idn, g, b))
    yr = torch.nn.functional.layer_norm(xr + ir, (C,), gr, br, 1e-5)
    (yr * cot).sum().backward()
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(y, yr, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ia.grad, ir.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xa.grad.float(), xr.grad, **tol)
    torch.testing.assert_close(ga.grad, gr.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(ba.grad, br.grad, rtol=1e-3, atol=1e-3)
    # train mode: x-gradient is zero exactly where the element was dropped; identity untouched
    xa2 = x.clone().float().requires_grad_()
    y2 = add_dropout_layernorm(xa2, torch.zeros_like(idn), torch.ones_like(g), torch.zeros_like(b),
                               0.25, training=True)
    y2.square().sum().backward()
    dropped = (xa2.grad == 0).float().mean().item()
    assert 0.22 < dropped < 0.28, dropped


def test_lowp_weight_shadows_follow_optimizer_steps():
    """The low-precision weight shadows (unibev_amd/linear.py) are refreshed on entry of every
    ``lowp_step_cache`` context: after an optimizer step the next forward sees the new weights
    (the fused AdamW kernel does not bump parameter versions, so nothing may be cached across
    steps), and gradients reach the f32 master parameters."""
    from unibev_amd.linear import linear, linear_cat, lowp_step_cache
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).to(DEV)
    lin2 = torch.nn.Linear(64, 16).to(DEV)
    opt = torch.optim.AdamW(list(lin.parameters()) + list(lin2.parameters()), lr=0.1, fused=True)
    x = torch.randn(4096, 64, device=DEV)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16), lowp_step_cache():
            y = linear(x, lin.weight, lin.bias)
            z = linear_cat(x, (lin.weight, lin2.weight), (lin.bias, lin2.bias))
        ref = torch.nn.functional.linear(x.bfloat16(), lin.weight.bfloat16(), lin.bias.bfloat16())
        torch.testing.assert_close(y, ref, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(z[:, :32], ref, rtol=2e-2, atol=2e-2)
        assert y.dtype == torch.bfloat16
        (y.float().square().mean() + z.float().square().mean()).backward()
        assert lin.weight.grad.dtype == torch.float32 and lin2.bias.grad is not None
        opt.step()
    # f32 path: gradients equal torch's
    a = torch.randn(8192, 64, device=DEV, requires_grad=True)
    w = torch.randn(32, 64, device=DEV, requires_grad=True)
    b = torch.randn(32, device=DEV, requires_grad=True)
    linear(a, w, b).square().sum().backward()
    ga, gw, gb = a.grad.clone(), w.grad.clone(), b.grad.clone()
    a.grad = w.grad = b.grad = None
    torch.nn.functional.linear(a, w, b).square().sum().backward()
    # split-bf16 products (~2^-17 each) in forward and input gradient: 1e-4 of the largest value
    torch.testing.assert_close(ga, a.grad, rtol=1e-4, atol=1e-4 * float(a.grad.abs().max()))
    torch.testing.assert_close(gw, w.grad, rtol=1e-4, atol=1e-4 * float(w.grad.abs().max()))
    torch.testing.assert_close(gb, b.grad, rtol=1e-4, atol=1e-4 * float(b.grad.abs().max()))
    # IEEE f32 GEMMs on request: equal to torch's to round-off
    from unibev_amd.linear import set_f32_gemm
    prev = set_f32_gemm('library')
    try:
        a.grad = w.grad = b.grad = None
        linear(a, w, b).square().sum().backward()
        torch.testing.assert_close(a.grad, ga, rtol=1e-4, atol=1e-4 * float(ga.abs().max()))
        gl = a.grad.clone()
        a.grad = w.grad = b.grad = None
        torch.nn.functional.linear(a, w, b).square().sum().backward()
        torch.testing.assert_close(gl, a.grad, rtol=2e-6, atol=2e-6 * float(a.grad.abs().max()))
    finally:
        set_f32_gemm(prev)


def test_two_forward_passes_before_backward():
    """Gradient accumulation style: two forward passes, then both backward passes.  The weight
    shadows must not be rewritten between them (the copies saved for backward would be invalidated),
    and an optimizer step or an in-place parameter edit must still refresh them."""
    from unibev_amd.linear import linear, lowp_step_cache
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).to(DEV)
    opt = torch.optim.SGD(lin.parameters(), lr=0.5)
    x = torch.randn(512, 64, device=DEV)

    def fwd():
        with torch.autocast('cuda', dtype=torch.bfloat16), lowp_step_cache():
            return linear(x, lin.weight, lin.bias).float().square().sum()

    fwd()                                   # creates the shadows
    a, b = fwd(), fwd()
    (a + b).backward()                      # raised "modified by an inplace operation" before
    g2 = lin.weight.grad.clone()
    lin.weight.grad = lin.bias.grad = None
    fwd().backward()
    torch.testing.assert_close(g2, 2 * lin.weight.grad, rtol=1e-3, atol=1e-3)
    before = fwd().item()
    opt.step()                              # optimizer step -> refresh on the next pass
    after = fwd().item()
    assert abs(after - before) > 1e-3 * abs(before)
    with torch.no_grad():
        lin.weight.mul_(0.0)                # in-place edit bumps the version counter -> refresh
        lin.bias.mul_(0.0)
    assert fwd().item() == 0.0


@pytest.mark.parametrize('dtype,fwd_tol,cos_lo,norm_tol', [(torch.float16, 1.5e-2, 0.95, 0.1),
                                                           (torch.bfloat16, 0.1, 0.75, 0.2)])
def test_fullsize_training_gradients_16bit_vs_fp32(dtype, fwd_tol, cos_lo, norm_tol):
    """cfg4 shapes (6 x 8x22 image feats, 180x180 LiDAR feats, 200x200 BEV, 3 layers), train-style
    backward with dropout off: the whole 16-bit path (16-bit residual stream, 16-bit offsets /
    logits, bins + MFMA owner tiles, fused norm / activation / Linear reductions) against the f32
    path — output normwise, gradients by direction and norm, parameter group by parameter group.
    The feature maps are box-filtered: on i.i.d. random pixels bilinear sampling turns the 2^-9
    rounding of a 16-bit offset into an O(1) output change and nothing meaningful can be compared
    (measured: 19 % output distance in bf16 on the raw maps, 5.9 % on the filtered ones; the f32 and
    16-bit residual streams are within 15 % of each other on this measure either way)."""
    import torch.nn.functional as F
    cfg, sd, inp, g = encoder_case('fullsize')
    model = _build(cfg).to(DEV).eval()                 # eval: dropout and modality dropout off
    _load(model, sd)
    torch.manual_seed(0)

    def smooth(x, k=5):
        y = F.avg_pool2d(x.reshape(-1, 1, *x.shape[-2:]), k, 1, k // 2, count_include_pad=False)
        return (y * k).reshape(x.shape)

    cot = None
    grads, outs = {}, {}
    for dt in (torch.float32, dtype):
        model.zero_grad(set_to_none=True)
        img = [smooth(t(x, torch.float32, DEV)).requires_grad_() for x in inp['img']]
        pts = [smooth(t(x, torch.float32, DEV)).requires_grad_() for x in inp['pts']]
        bev_q = t(inp['bev_q'], torch.float32, DEV).requires_grad_()
        with torch.autocast('cuda', dtype=dt, enabled=dt != torch.float32):
            fused = model.encode(img, pts, bev_q, inp['bev_h'], inp['bev_w'],
                                 bev_pos=t(inp['bev_pos'], torch.float32, DEV), img_metas=inp['metas'])
        if cot is None:
            cot = torch.randn_like(fused.float()) / fused.shape[0] ** 0.5
        (fused.float() * cot).sum().backward()
        outs[dt] = fused.detach().float()
        gd = {'bev_q': bev_q.grad, 'img': img[0].grad, 'pts': pts[0].grad}
        for n, p in model.named_parameters():
            if p.grad is not None:
                parts = n.split('.')
                gd.setdefault(parts[0] + ':' + '.'.join(parts[-2:]), []).append(p.grad.flatten().float())
        grads[dt] = {k: (torch.cat(v) if isinstance(v, list) else v.flatten().float()) for k, v in gd.items()}
    ferr = float((outs[dtype] - outs[torch.float32]).norm() / outs[torch.float32].norm())
    assert ferr < fwd_tol, ferr
    checked = 0
    for k, a in grads[torch.float32].items():
        b = grads[dtype][k]
        if float(a.norm()) == 0.0:
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        ratio = float(b.norm() / a.norm())
        assert cos > cos_lo and abs(ratio - 1.0) < norm_tol, (k, cos, ratio)
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize('gemm,bar', [('library', 2.5e-3), ('mfma', 4e-3)])
def test_fullsize_gradient_parity_statement(gemm, bar):
    """The gradient-parity STATEMENT at BASELINE's shapes: fixture ``fullsize_smooth`` — box-filtered maps and sampling
    parameters shifted off the pixel centres (synthetic.smooth_like_state_dict), recorded from the reference — is well
    conditioned (no sample on a kink of the bilinear interpolant, d/d(location) nearly continuous), so EVERY tensor is
    held to ONE normwise bar with no exclusions and no allowance for 1-D parameters: the inputs, the BEV queries, every
    encoder-side parameter including the self-attention ``sampling_offsets``.

    The bars are the f32 reference's own round-off floor: the oracle (the reference's arithmetic) run in float32 sits
    1.2e-3 (median over the tensors) to 2.7e-3 (worst: a layer-0 attention_weights bias) from its float64 run on this
    fixture (tools/oracle_f32_floor.py, profiles/r04_gradient_floor.txt) — three layers of sampling amplify a unit
    round-off ~1000x whatever computes them.  Measured on MI355X against the f32 oracle: IEEE library GEMMs worst
    2.0e-3 (forward 1.3e-6), the default split-bf16 MFMA Linear layers worst 3.1e-3; against the float64 oracle the
    library path is 3.0e-3 at worst, i.e. as far from exact as the reference is."""
    from unibev_amd.linear import set_f32_gemm
    prev = set_f32_gemm(gemm)
    try:
        _fullsize_gradients('fullsize_smooth', norm_bar=bar, elem_bar=0.1, fwd_bar=1e-4, allow_1d=1.0, skip_kinks=False)
    finally:
        set_f32_gemm(prev)


@pytest.mark.parametrize('fixture', ['fullsize_init', 'fullsize'])
def test_fullsize_gradients_vs_oracle(fixture):
    """BASELINE shapes (200x200 BEV, 6 x 8x22 image maps, 180x180 LiDAR map, C = 256, 3 layers, CNW): gradient of
    sum(fused * cot) w.r.t. the inputs, the BEV queries and every encoder-side parameter, f32 product path against
    torch autograd through the CPU oracle (the small fixtures were the only gradient parity).

    Gradients are worse conditioned than the forward: measured on MI355X, `fullsize_init` (the bench's operating
    point) forward 7e-6, gradients 2.0 - 2.7e-3 normwise per tensor (1.1e-3 with IEEE library GEMMs, so half of it is
    the split-bf16 products' 2^-17 amplified through three layers); `fullsize` (i.i.d. maps, random offset weights —
    the adversarial fixture of DESIGN.md section 4) forward 3.1e-4, gradients 4 - 4.6e-2.  Bars: 2x the measured.
    At the INITIAL sampling parameters the self-attention's offsets are whole pixels from pixel-centred reference
    points: every sample sits exactly on a kink of the bilinear interpolant, where d / d(location) jumps between
    its one-sided values and a 1-ulp difference in the location picks the side — the self-attention
    `sampling_offsets` gradients of that fixture (0.3 - 0.5 apart, with library GEMMs too) are left out; on the
    random-parameter fixture they agree like every other tensor."""
    NORM_BAR, ELEM_BAR = (5e-3, 6e-2) if fixture == 'fullsize_init' else (9e-2, 0.7)
    _fullsize_gradients(fixture, NORM_BAR, ELEM_BAR, 1e-3, allow_1d=1.0, skip_kinks=fixture == 'fullsize_init')


def _fullsize_gradients(fixture, norm_bar, elem_bar, fwd_bar, allow_1d, skip_kinks, oracle_dtype=torch.float32):
    from oracle import unibev_ref as R
    NORM_BAR, ELEM_BAR, FWD_BAR = norm_bar, elem_bar, fwd_bar
    cfg, sd, inp, g = encoder_case(fixture)
    nq, bs, width = inp['bev_h'] * inp['bev_w'], inp['bs'], cfg['embed_dims'] * (2 if cfg.get('fusion_method') == 'cat' else 1)
    cot = syn.seeded_array('cot:' + fixture, (nq, bs, width), 5) / nq ** 0.5
    torch.set_num_threads(min(16, torch.get_num_threads()))
    od = oracle_dtype
    P = {k: v.to(od).requires_grad_() for k, v in R.state_dict_to_torch(sd).items()}
    oi = [t(x, od).requires_grad_() for x in inp['img']]
    op = [t(x, od).requires_grad_() for x in inp['pts']]
    oq = t(inp['bev_q'], od).requires_grad_()
    fused_ref = R.transformer_encode_fuse(P, cfg, oi, op, oq, inp['bev_h'], inp['bev_w'], t(inp['bev_pos'], od), inp['metas'])
    (fused_ref * t(cot, od)).sum().backward()
    model = _build(cfg).to(DEV).eval()
    _load(model, sd)
    gi = [t(x, device=DEV).requires_grad_() for x in inp['img']]
    gp = [t(x, device=DEV).requires_grad_() for x in inp['pts']]
    gq = tq(inp['bev_q'], device=DEV, grad=True)
    fused = model.encode(gi, gp, gq, inp['bev_h'], inp['bev_w'], bev_pos=t(inp['bev_pos'], device=DEV),
                         img_metas=inp['metas'])
    (fused * t(cot, device=DEV)).sum().backward()
    ferr = float((fused.detach().cpu().to(od) - fused_ref.detach()).norm() / fused_ref.detach().norm())
    assert ferr < FWD_BAR, ferr
    worst = {}

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach().double()
        nb = float(b.norm())
        if nb == 0.0:
            assert float(a.norm()) == 0.0, what
            return
        nerr = float((a - b).norm()) / nb
        merr = float((a - b).abs().max()) / float(b.abs().max())
        # 1-D parameters (biases, norm scales) are sums of ~80 000 signed terms accumulated with f32 atomics in
        # arrival order: their distance moves from run to run (one sampling_offsets bias: 2.0e-3 .. 5.9e-3 over 8 runs)
        worst[what] = (nerr / (allow_1d if b.dim() == 1 else 1.0), merr)
    close(gi[0].grad, oi[0].grad, 'img feats')
    close(gp[0].grad, op[0].grad, 'pts feats')
    close(gq.grad, oq.grad, 'bev queries')
    checked = 0
    for k, p in model.named_parameters():
        if k.startswith('decoder') or k.startswith('reference_points') or P[k].grad is None:
            continue
        assert p.grad is not None, k
        if skip_kinks and '.attentions.0.sampling_offsets.' in k:
            continue                                        # samples on the interpolant's kinks, see above
        close(p.grad, P[k].grad, k)
        checked += 1
    assert checked > 100, checked
    w = max(worst.items(), key=lambda kv: kv[1][0])
    print(f'fullsize gradients: forward {ferr:.1e}, worst normwise {w[0]} {w[1][0]:.1e}, worst element '
          f'{max(v[1] for v in worst.values()):.1e}')
    for k, v in sorted(worst.items(), key=lambda kv: -kv[1][0])[:int(__import__('os').environ.get('UBV_TEST_SHOW', '12'))]:
        print(f'   {k:70s} {v[0]:.2e} {v[1]:.2e}')
    bad = {k: v for k, v in worst.items() if not (v[0] < NORM_BAR and v[1] < ELEM_BAR)}
    assert not bad, bad


def test_first_self_attention_shared_across_the_batch_equals_per_sample():
    """The first encoder layer's queries are one table for every sample, so its self-attention (projections, sampling,
    output_proj) runs for ONE sample and the samples part at the dropout of the fused add + LayerNorm (row-period reads,
    ``bcast_rows``).  Training mode with dropout on, same seeds: the BEV features equal the per-sample run bit for bit
    (every row sees the same arithmetic and the same mask) and every gradient agrees to f32 round-off (the batch sum of
    the shared rows' gradients moves from behind the GEMMs to in front of them)."""
    from unibev_amd.modules import encoders as E
    cfg, sd, inp, g = encoder_case('cnw')
    model = _build(cfg).to(DEV).train()
    _load(model, sd)
    model.forced_flags = (1, 1)
    img = [t(x, torch.float32, DEV).requires_grad_() for x in inp['img']]
    pts = [t(x, torch.float32, DEV).requires_grad_() for x in inp['pts']]
    bev_q = tq(inp['bev_q'], torch.float32, DEV, grad=True)
    bs = inp['bs']
    assert bs > 1
    bev_pos = t(inp['bev_pos'], torch.float32, DEV)[:1].expand(bs, -1, -1, -1)  # one positional table, as the model's
    params = [p for n, p in model.named_parameters() if p.requires_grad and not n.startswith('reference_points')]
    cot = torch.randn(inp['bev_h'] * inp['bev_w'], bs, g['fused'].shape[-1], device=DEV,
                      generator=torch.Generator(DEV).manual_seed(3))
    calls = []
    orig = E.UF.add_dropout_layernorm

    def spy(*a, **k):
        calls.append(int(k.get('batch', 0) or 0))
        return orig(*a, **k)

    def run(share):
        E._SHARE_FIRST = share
        torch.manual_seed(11)                                   # the dropout seeds come from torch's CPU generator
        for x in params + img + pts + [bev_q]:
            x.grad = None
        out = model.encode(img, pts, bev_q, inp['bev_h'], inp['bev_w'], bev_pos=bev_pos, img_metas=inp['metas'])
        (out * cot).sum().backward()
        return out.detach().clone(), [None if x.grad is None else x.grad.clone() for x in params + img + pts + [bev_q]]

    try:
        E.UF.add_dropout_layernorm = spy
        out1, g1 = run(True)
        shared_calls = list(calls)
        calls.clear()
        out0, g0 = run(False)
    finally:
        E.UF.add_dropout_layernorm = orig
        E._SHARE_FIRST = True
    assert shared_calls.count(bs) == 2 and calls.count(bs) == 0, (shared_calls, calls)   # one per encoder, first layer
    assert torch.equal(out1, out0)
    for a, b in zip(g1, g0):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).norm()) <= 2e-5 * float(b.norm()) + 1e-12
