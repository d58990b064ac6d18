"""GridMask oracle (oracle/grid_mask_ref.py) against the masks and the RNG protocol recorded from the reference
(tests/golden/grid_mask.npz, made by tests/golden/make_golden.py::gen_grid_mask) — no GPU."""
import os

import numpy as np
import pytest

from oracle import grid_mask_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'grid_mask.npz'))
SEEDS = sorted(int(k[1:].split('_')[0]) for k in GOLD.files if k.startswith('s') and k.endswith('_mask'))


@pytest.mark.parametrize('seed', SEEDS)
def test_oracle_reproduces_the_recorded_mask_and_draw_order(seed):
    n, c, h, w, prob = GOLD[f's{seed}_meta']
    np.random.seed(seed)
    drawn = R.draw(int(h), prob / 100.0)
    m = np.ones((h, w), np.float32) if drawn is None else R.mask(int(h), int(w), *drawn)
    assert np.array_equal(m.astype(np.uint8), GOLD[f's{seed}_mask'])
    assert np.random.rand() == GOLD[f's{seed}_next'][0]          # consumed exactly the reference's draws


VSEEDS = sorted(int(k[1:].split('_')[0]) for k in GOLD.files if k.startswith('v') and k.endswith('_mask'))


@pytest.mark.parametrize('seed', VSEEDS)
def test_oracle_reproduces_rotated_and_filled_masks(seed):
    """rotate > 1 (the restated PIL nearest-neighbour rotation) and offset=True against the reference's recordings."""
    n, c, h, w, rotate, offset, mode = (int(v) for v in GOLD[f'v{seed}_meta'])
    np.random.seed(seed)
    drawn = R.draw(h, 1.0, rotate=rotate)
    m = R.mask(h, w, *drawn, mode=mode)
    assert np.array_equal(m.astype(np.uint8), GOLD[f'v{seed}_mask'])
    if offset:
        assert np.array_equal(R.fill_draw(h, w) * (1 - m), GOLD[f'v{seed}_fill'])
    else:
        assert not GOLD[f'v{seed}_fill'].any()
    assert np.random.rand() == GOLD[f'v{seed}_next'][0]


def test_restated_rotation_equals_pillow_where_pillow_is_installed():
    Image = pytest.importorskip('PIL.Image')
    rs = np.random.RandomState(3)
    for _ in range(120):
        h, w = int(rs.randint(4, 80)), int(rs.randint(4, 100))
        img = (rs.rand(h, w) > 0.5).astype(np.uint8)
        for ang in (int(rs.randint(0, 360)), 90, 180, 270):
            assert np.array_equal(np.asarray(Image.fromarray(img).rotate(ang)), R.rotate_nearest(img, ang)), (h, w, ang)


def test_fixture_covers_the_skip_branch_and_the_masked_branch():
    kept = [float(GOLD[f's{s}_mask'].mean()) for s in SEEDS]
    assert any(k == 1.0 for k in kept) and any(0.0 < k < 1.0 for k in kept)


def test_mode0_is_the_complement():
    a = R.mask(20, 50, 7, 4, 3, 5, mode=1)
    b = R.mask(20, 50, 7, 4, 3, 5, mode=0)
    assert np.array_equal(a + b, np.ones_like(a))


def test_module_keeps_the_reference_constructor_and_rejects_what_is_not_built():
    from unibev_amd.modules.grid_mask import GridMask
    gm = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)
    gm.set_prob(3, 6)
    assert gm.prob == pytest.approx(0.35) and gm.st_prob == 0.7
    assert GridMask(True, True, rotate=45, offset=True).rotate == 45
    import torch
    x = torch.ones(1, 1, 8, 8)
    assert gm.eval()(x) is x                                      # evaluation: untouched, as the reference
    np.random.seed(4)
    gm.prob = 0.0
    assert gm.train()(x) is x                                     # skipped pass never reaches the device
