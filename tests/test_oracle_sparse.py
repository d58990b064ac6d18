"""Known-answer checks of the sparse-convolution oracle (oracle/sparse_conv_ref.py): the published spconv
semantics on hand-built inputs (no GPU)."""
import torch

from oracle import sparse_conv_ref as R


def test_submanifold_keeps_the_active_set_and_ignores_inactive_neighbours():
    coors = torch.tensor([[0, 1, 1, 1], [0, 1, 1, 2], [0, 3, 3, 3]], dtype=torch.int32)
    feats = torch.tensor([[1.0], [10.0], [100.0]])
    w = torch.zeros(3, 3, 3, 1, 1)
    w[1, 1, 1] = 1.0          # centre
    w[1, 1, 2] = 0.5          # +x neighbour
    dense, mask = R.densify(feats, coors, 1, (5, 5, 5))
    out, m = R.subm_conv(dense, mask, w)
    assert torch.equal(m, mask)
    assert out[0, 0, 1, 1, 1] == 1.0 + 0.5 * 10.0 and out[0, 0, 1, 1, 2] == 10.0 and out[0, 0, 3, 3, 3] == 100.0
    assert out.sum() == 6.0 + 10.0 + 100.0        # nothing leaks to inactive sites


def test_strided_conv_activates_every_site_whose_field_holds_an_input():
    coors = torch.tensor([[0, 2, 2, 2]], dtype=torch.int32)
    feats = torch.tensor([[2.0]])
    w = torch.ones(3, 3, 3, 1, 1)
    dense, mask = R.densify(feats, coors, 1, (5, 5, 5))
    out, m = R.sparse_conv(dense, mask, w, stride=(2, 2, 2), padding=(1, 1, 1))
    # out = floor((5 + 2 - 3) / 2) + 1 = 3 per axis; input 2 is covered by outputs o with 2 o - 1 <= 2 <= 2 o + 1: o = 1 only
    assert out.shape == (1, 1, 3, 3, 3)
    assert m.sum() == 1 and out[0, 0, 1, 1, 1] == 2.0
    coors = torch.tensor([[0, 1, 1, 1]], dtype=torch.int32)       # odd coordinate: two outputs per axis
    dense, mask = R.densify(feats, coors, 1, (5, 5, 5))
    out, m = R.sparse_conv(dense, mask, w, stride=(2, 2, 2), padding=(1, 1, 1))
    assert m.sum() == 8 and float(out.sum()) == 16.0


def test_batch_norm_uses_only_active_sites():
    coors = torch.tensor([[0, 0, 0, 0], [0, 1, 1, 1]], dtype=torch.int32)
    feats = torch.tensor([[1.0], [3.0]])
    dense, mask = R.densify(feats, coors, 1, (2, 2, 2))
    out = R.batch_norm(dense, mask, torch.ones(1), torch.zeros(1), 0.0)
    assert torch.allclose(out[0, 0, 0, 0, 0], torch.tensor(-1.0)) and torch.allclose(out[0, 0, 1, 1, 1], torch.tensor(1.0))
    assert float(out.abs().sum()) == 2.0
