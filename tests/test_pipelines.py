"""Host data contract: the pipeline stages that produce the two meta keys
the encoder reads, and the one-upload-per-batch form of ``lidar2img``."""
import numpy as np
import pytest
import torch

from unibev_amd import synthetic as syn
from unibev_amd.pipelines import (PIPELINES, CustomCollect3D, NormalizeMultiviewImage, PadMultiViewImage,
                                  metas_to_device)


def _views(n=6, h=900, w=1600, seed=0):
    rs = np.random.RandomState(seed)
    return [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(n)]


def test_pad_multiview_like_the_configs():
    """size_divisor=32 turns 900x1600 views into 928x1600 (configs/unibev: img_shape the encoder
    divides by); fixed size pads bottom/right with pad_val."""
    res = PIPELINES.build(dict(type='PadMultiViewImage', size_divisor=32))(dict(img=_views(2)))
    assert res['img_shape'] == [(928, 1600, 3)] * 2 == res['pad_shape'] and res['ori_shape'] == [(900, 1600, 3)] * 2
    assert res['pad_size_divisor'] == 32 and res['pad_fixed_size'] is None
    assert (res['img'][0][900:] == 0).all() and res['img'][0].dtype == np.uint8
    src = _views(1, 5, 7)
    res = PadMultiViewImage(size=(8, 9), pad_val=3)(dict(img=[src[0].copy()]))
    assert res['img'][0].shape == (8, 9, 3) and (res['img'][0][5:] == 3).all() and (res['img'][0][:, 7:] == 3).all()
    np.testing.assert_array_equal(res['img'][0][:5, :7], src[0])


def test_normalize_multiview():
    mean, std = [103.530, 116.280, 123.675], [1.0, 1.0, 1.0]
    src = _views(2, 4, 5)
    res = NormalizeMultiviewImage(mean, std, to_rgb=False)(dict(img=[v.copy() for v in src]))
    assert res['img'][0].dtype == np.float32 and res['img_norm_cfg']['to_rgb'] is False
    np.testing.assert_allclose(res['img'][1], src[1].astype(np.float32) - np.float32(mean), rtol=0, atol=1e-4)
    res = NormalizeMultiviewImage([1, 2, 3], [2, 4, 8], to_rgb=True)(dict(img=[src[0].copy()]))
    exp = (src[0][..., ::-1].astype(np.float32) - np.float32([1, 2, 3])) / np.float32([2, 4, 8])
    np.testing.assert_allclose(res['img'][0], exp, rtol=1e-6)


def test_collect_and_device_metas():
    metas = syn.img_metas(2, 6, (256, 704), jitter_seed=1)
    res = dict(img='IMG', points='PTS', lidar2img=metas[0]['lidar2img'], img_shape=metas[0]['img_shape'],
               unrelated=1)
    data = CustomCollect3D(keys=['img', 'points'])(res)
    assert set(data) == {'img', 'points', 'img_metas'} and set(data['img_metas']) == {'lidar2img', 'img_shape'}
    dev = metas_to_device(metas, 'cpu')
    assert all(torch.is_tensor(m['lidar2img']) and m['lidar2img'].shape == (6, 4, 4) and
               m['lidar2img'].dtype == torch.float32 for m in dev)
    assert dev[0]['lidar2img'].untyped_storage().data_ptr() == dev[1]['lidar2img'].untyped_storage().data_ptr()
    np.testing.assert_array_equal(dev[1]['lidar2img'].numpy(), np.asarray(metas[1]['lidar2img']).astype(np.float32))
    assert dev[0]['img_shape'] == metas[0]['img_shape']
    # the encoder takes either form and reads the same numbers
    from unibev_amd.modules.encoders import _lidar2img_tensor
    a = _lidar2img_tensor(metas, torch.device('cpu'))
    b = _lidar2img_tensor(dev, torch.device('cpu'))
    assert torch.equal(a, b) and _lidar2img_tensor(metas, torch.device('cpu')) is a      # cached upload


def test_pipeline_stages_vs_reference_recorded_vectors():
    """tests/golden/pipelines.npz: the reference's own ``NormalizeMultiviewImage`` -> ``PadMultiViewImage`` ->
    ``CustomCollect3D`` (transform_3d.py:7-95, 199-284, imported by tests/golden/make_golden.py) on seeded views, in
    the shipped configs' order and in an RGB / fixed-size variant: pixel values, every shape key, the collected key
    sets and their order, the norm cfg and the reprs."""
    import json
    from _util import golden, checksum
    import make_golden as mg
    g = golden('pipelines')
    views = mg.pipeline_views()
    np.testing.assert_array_equal(checksum(np.stack(views)), g['views_ck'])
    metas = syn.img_metas(1, mg.PIPELINE_VIEWS[0], mg.PIPELINE_VIEWS[1:3])[0]
    for tag, norm, pad in (('cfg', dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False),
                            dict(size_divisor=32)),
                           ('rgb', dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
                            dict(size=(40, 56), pad_val=2))):
        res = dict(img=[v.copy() for v in views], lidar2img=metas['lidar2img'], sample_idx='tok', unrelated=1,
                   points='PTS', pts_filename='a.bin', box_type_3d='LiDAR')
        res = PIPELINES.build(dict(type='NormalizeMultiviewImage', **norm))(res)
        res = PIPELINES.build(dict(type='PadMultiViewImage', **pad))(res)
        data = PIPELINES.build(dict(type='CustomCollect3D', keys=['points', 'img']))(res)
        assert list(data) == json.loads(str(g[tag + '_data_keys']))
        m = data['img_metas']
        assert list(m) == json.loads(str(g[tag + '_meta_keys']))
        got = np.stack(data['img'])
        assert got.dtype == g[tag + '_img'].dtype and got.shape == g[tag + '_img'].shape
        np.testing.assert_array_equal(got, g[tag + '_img'])                 # bit-exact: same f32 operations
        for k in ('img_shape', 'pad_shape', 'ori_shape'):
            np.testing.assert_array_equal(np.asarray(m[k]), g[f'{tag}_{k}'])
        np.testing.assert_array_equal(m['img_norm_cfg']['mean'], g[tag + '_norm_mean'])
        np.testing.assert_array_equal(m['img_norm_cfg']['std'], g[tag + '_norm_std'])
        assert int(m['img_norm_cfg']['to_rgb']) == int(g[tag + '_norm_to_rgb'])
        fixed, div = g[tag + '_pad_fixed_size'], int(g[tag + '_pad_size_divisor'])
        assert (res['pad_fixed_size'] is None and fixed.tolist() == [-1]) or list(res['pad_fixed_size']) == fixed.tolist()
        assert (res['pad_size_divisor'] is None and div == -1) or res['pad_size_divisor'] == div
        reprs = json.loads(str(g[tag + '_repr']))
        assert repr(PadMultiViewImage(**pad)) == reprs[0]
        assert repr(CustomCollect3D(keys=['img'])) == reprs[1]


@pytest.mark.gpu
def test_metas_to_device_feeds_point_sampling_on_the_gpu():
    """One pinned upload per batch; the encoder projects from the device views and gets the visibility / camera
    coordinates it gets from host arrays (reference: encoder_unibev_detr_img.py:115-124 rebuilds the tensor from numpy
    on every forward)."""
    from unibev_amd.modules.encoders import ImgEncoder
    dev = torch.device('cuda')
    metas = syn.img_metas(2, 6, (256, 704), jitter_seed=2)
    on_dev = metas_to_device(metas, dev)
    assert all(m['lidar2img'].is_cuda and m['lidar2img'].dtype == torch.float32 for m in on_dev)
    assert on_dev[0]['lidar2img'].untyped_storage().data_ptr() == on_dev[1]['lidar2img'].untyped_storage().data_ptr()
    ref3d = ImgEncoder.get_reference_points(50, 50, 8, 4, dim='3d', bs=2, device=dev)
    enc = ImgEncoder.__new__(ImgEncoder)
    pc = [-54, -54, -5, 54, 54, 3]
    cam_a, mask_a = ImgEncoder.point_sampling(enc, ref3d, pc, metas)
    cam_b, mask_b = ImgEncoder.point_sampling(enc, ref3d, pc, on_dev)
    assert torch.equal(mask_a, mask_b) and torch.equal(cam_a, cam_b) and mask_a.any()
