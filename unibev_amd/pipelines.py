"""Host-side data contract of the hot path.

The encoder reads exactly two things from ``img_metas`` — ``lidar2img`` (Nc, 4, 4) and
``img_shape`` [(H, W, 3)] * Nc (encoder_unibev_detr_img.py:115-118, 166-167) — and they are
produced by three pipeline stages of the reference
(projects/UniBEV/unibev_plugin/datasets/pipelines/transform_3d.py):

  ``PadMultiViewImage``        :7-57    pads every view bottom/right, writes ``img_shape`` /
                                        ``pad_shape`` / ``ori_shape`` (the 900 -> 928 rows of the configs)
  ``NormalizeMultiviewImage``  :60-95   (BGR -> RGB,) subtract mean, divide by std, float32
  ``CustomCollect3D``          :199-284 gathers ``meta_keys`` into ``img_metas``

restated here on numpy alone ([ext] mmcv.impad / impad_to_multiple / imnormalize behaviour), same
registry key / constructor kwargs / result-dict keys, plus what the GPU path wants from the loader:
``metas_to_device`` uploads a batch's ``lidar2img`` ONCE (one (B, Nc, 4, 4) tensor; every meta then
holds a device view), so ``ImgEncoder`` projects without a host-to-device copy per forward pass.
"""
import numpy as np
import torch

from .registry import Registry

PIPELINES = Registry('pipeline')


def _impad(img, shape, pad_val=0):
    """[ext] mmcv.impad(img, shape=...): zero (pad_val) padding on the bottom and right."""
    h, w = int(shape[0]), int(shape[1])
    assert h >= img.shape[0] and w >= img.shape[1], 'pad target smaller than the image'
    out = np.full((h, w) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


@PIPELINES.register_module()
class PadMultiViewImage:
    """Pad every view to a fixed ``size`` (h, w) or up to a multiple of ``size_divisor``."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert size is not None or size_divisor is not None
        assert size is None or size_divisor is None
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val

    def __call__(self, results):
        if self.size is not None:
            padded = [_impad(img, self.size, self.pad_val) for img in results['img']]
        else:
            d = self.size_divisor
            padded = [_impad(img, (-(-img.shape[0] // d) * d, -(-img.shape[1] // d) * d), self.pad_val)
                      for img in results['img']]
        results['ori_shape'] = [img.shape for img in results['img']]
        results['img'] = padded
        results['img_shape'] = [img.shape for img in padded]
        results['pad_shape'] = [img.shape for img in padded]
        results['pad_fixed_size'] = self.size
        results['pad_size_divisor'] = self.size_divisor
        return results

    def __repr__(self):
        return (f'{self.__class__.__name__}(size={self.size}, size_divisor={self.size_divisor}, '
                f'pad_val={self.pad_val})')


@PIPELINES.register_module()
class NormalizeMultiviewImage:
    """(img[..., ::-1] if to_rgb) -> float32 -> (img - mean) / std per channel ([ext]
    mmcv.imnormalize: subtract, then multiply by the reciprocal of std)."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb

    def __call__(self, results):
        stdinv = (1.0 / np.float64(self.std.reshape(1, -1))).astype(np.float32)
        out = []
        for img in results['img']:
            x = np.asarray(img, dtype=np.float32)
            if self.to_rgb:
                x = x[..., ::-1]
            out.append(np.ascontiguousarray((x - self.mean.reshape(1, -1)) * stdinv))
        results['img'] = out
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})'


@PIPELINES.register_module()
class CustomCollect3D:
    """Collect ``keys`` and the ``meta_keys`` present in the results ([ext] DataContainer wrapping
    is a transport detail of mmcv's collate; ``img_metas`` is the plain dict here)."""

    DEFAULT_META = ('filename', 'ori_shape', 'img_shape', 'lidar2img', 'depth2img', 'cam2img',
                    'pad_shape', 'scale_factor', 'flip', 'pcd_horizontal_flip', 'pcd_vertical_flip',
                    'box_mode_3d', 'box_type_3d', 'img_norm_cfg', 'pcd_trans', 'sample_idx',
                    'prev_idx', 'next_idx', 'pcd_scale_factor', 'pcd_rotation', 'pts_filename',
                    'transformation_3d_flow', 'scene_token', 'can_bus')

    def __init__(self, keys, meta_keys=DEFAULT_META):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_metas': {k: results[k] for k in self.meta_keys if k in results}}
        for k in self.keys:
            data[k] = results[k]
        return data

    def __repr__(self):
        return f'{self.__class__.__name__}(keys={self.keys}, meta_keys={self.meta_keys})'


def metas_to_device(img_metas, device, non_blocking=True):
    """Upload the batch's projection matrices once: ``lidar2img`` of every sample becomes a view of
    one (B, Nc, 4, 4) float32 tensor on ``device`` (float64 arrays are rounded to f32 exactly as
    ``reference_points.new_tensor`` does, encoder_unibev_detr_img.py:121-124).  Returns new meta
    dicts; everything else is passed through."""
    arr = np.ascontiguousarray(np.asarray([m['lidar2img'] for m in img_metas]).astype(np.float32))
    host = torch.from_numpy(arr)
    if torch.device(device).type == 'cuda':
        host = host.pin_memory()
    l2i = host.to(device, non_blocking=non_blocking)
    return [dict(m, lidar2img=l2i[b]) for b, m in enumerate(img_metas)]
