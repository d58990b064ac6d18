"""Host data contract: the pipeline stages that produce the two meta keys
the encoder reads, and the one-upload-per-batch form of ``lidar2img``."""
import numpy as np
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
