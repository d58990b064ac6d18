"""Known-answer checks of the DCNv2 oracle (oracle/dcn_ref.py; mmcv is not in the reference tree: unpinned)."""
import torch
import torch.nn.functional as F

from oracle import dcn_ref as R


def test_zero_offsets_and_unit_mask_are_a_plain_convolution():
    torch.manual_seed(0)
    for (stride, pad, dil) in ((1, 1, 1), (2, 1, 1), (1, 2, 2)):
        x = torch.randn(2, 8, 9, 11)
        w = torch.randn(6, 8, 3, 3)
        b = torch.randn(6)
        Ho = (9 + 2 * pad - (dil * 2 + 1)) // stride + 1
        Wo = (11 + 2 * pad - (dil * 2 + 1)) // stride + 1
        off = torch.zeros(2, 18, Ho, Wo)
        m = torch.ones(2, 9, Ho, Wo)
        got = R.modulated_deform_conv2d(x, off, m, w, b, stride, pad, dil)
        assert torch.allclose(got, F.conv2d(x, w, b, stride, pad, dil), atol=1e-5)


def test_integer_offsets_shift_the_taps_and_the_mask_scales_them():
    x = torch.arange(25.0).view(1, 1, 5, 5)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0                                                   # centre tap only
    off = torch.zeros(1, 18, 5, 5)
    off[0, 2 * 4] = 1.0                                                   # centre tap: dy = +1
    off[0, 2 * 4 + 1] = -2.0                                              # dx = -2
    m = torch.full((1, 9, 5, 5), 0.5)
    got = R.modulated_deform_conv2d(x, off, m, w, None, 1, 1, 1)
    want = torch.zeros(5, 5)
    want[:4, 2:] = 0.5 * x[0, 0, 1:, :3]                                  # rows beyond the map read 0
    assert torch.equal(got[0, 0], want)


def test_half_pixel_offset_is_the_bilinear_mean_and_the_border_rule_holds():
    x = torch.ones(1, 1, 4, 4)
    w = torch.zeros(1, 1, 1, 1) + 1.0
    off = torch.zeros(1, 2, 4, 4)
    off[0, 0] = -0.5                                                      # half a pixel up: the row above the map reads 0
    m = torch.ones(1, 1, 4, 4)
    got = R.modulated_deform_conv2d(x, off, m, w, None, 1, 0, 1)
    assert torch.allclose(got[0, 0, 0], torch.full((4,), 0.5)) and torch.allclose(got[0, 0, 1:], torch.ones(3, 4))
    off[0, 0] = -1.0                                                      # exactly -1: outside, contributes nothing
    got = R.modulated_deform_conv2d(x, off, m, w, None, 1, 0, 1)
    assert torch.equal(got[0, 0, 0], torch.zeros(4))


def test_pack_splits_the_offset_convolution_like_mmcv():
    torch.manual_seed(1)
    x = torch.randn(1, 4, 6, 6)
    cw, cb = torch.zeros(27, 4, 3, 3), torch.zeros(27)                     # mmcv's init: zero offsets, mask = 0.5
    w = torch.randn(5, 4, 3, 3)
    got = R.dcn_pack(x, cw, cb, w, None, 1, 1, 1, 1)
    assert torch.allclose(got, 0.5 * F.conv2d(x, w, None, 1, 1), atol=1e-5)
