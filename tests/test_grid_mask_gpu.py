"""``ubv_grid_mask`` / ``modules.GridMask`` on the GPU against the reference-recorded masks (tests/golden/grid_mask.npz)
and the oracle (oracle/grid_mask_ref.py).  Bit-exact: the op multiplies by 0 or 1."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'grid_mask.npz'))
SEEDS = sorted(int(k[1:].split('_')[0]) for k in GOLD.files if k.startswith('s') and k.endswith('_mask'))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('seed', SEEDS)
def test_module_matches_the_recorded_reference_masks(seed, dtype):
    from unibev_amd.modules import GridMask
    n, c, h, w, prob = (int(v) for v in GOLD[f's{seed}_meta'])
    gm = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=prob / 100.0).train()
    x = torch.randn(n, c, h, w, device=DEV).to(dtype).requires_grad_(True)
    np.random.seed(seed)
    y = gm(x)
    assert np.random.rand() == GOLD[f's{seed}_next'][0]
    m = torch.from_numpy(GOLD[f's{seed}_mask']).to(DEV).to(dtype)
    assert torch.equal(y, x.detach() * m)
    g = torch.randn_like(y)
    if y.requires_grad and y is not x:
        y.backward(g)
        assert torch.equal(x.grad, g * m)


VSEEDS = sorted(int(k[1:].split('_')[0]) for k in GOLD.files if k.startswith('v') and k.endswith('_mask'))


@pytest.mark.parametrize('seed', VSEEDS)
def test_module_matches_the_recorded_rotated_and_filled_passes(seed):
    """rotate > 1 / offset=True (no shipped config): mask, fill and RNG protocol recorded from the reference."""
    from _util import checksum
    from unibev_amd import synthetic as syn
    from unibev_amd.modules import GridMask
    n, c, h, w, rotate, offset, mode = (int(v) for v in GOLD[f'v{seed}_meta'])
    gm = GridMask(True, True, rotate=rotate, offset=bool(offset), ratio=0.5, mode=mode, prob=1.0).train()
    x = torch.from_numpy(syn.seeded_array(f'grid_mask:v{seed}', (n, c, h, w), seed)).to(DEV).requires_grad_(True)
    np.random.seed(seed)
    y = gm(x)
    assert np.random.rand() == GOLD[f'v{seed}_next'][0]
    m = torch.from_numpy(GOLD[f'v{seed}_mask']).to(DEV).float()
    fill = torch.from_numpy(GOLD[f'v{seed}_fill']).to(DEV)
    assert torch.equal(y, x.detach() * m + fill)
    np.testing.assert_allclose(checksum(y.detach().cpu().numpy()), GOLD[f'v{seed}_y_ck'], rtol=1e-12)
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad, g * m)


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('use', [(True, True), (True, False), (False, True)])
def test_kernel_matches_the_oracle_over_random_geometries(use, mode):
    from oracle import grid_mask_ref as R
    import unibev_amd.functional as UF
    rs = np.random.RandomState(11)
    for _ in range(40):
        h, w = int(rs.randint(3, 70)), int(rs.randint(3, 90))
        d = int(rs.randint(2, h))
        length = min(max(int(d * rs.uniform(0.1, 0.9) + 0.5), 1), d - 1)
        st_h, st_w = int(rs.randint(d)), int(rs.randint(d))
        x = torch.randn(3, h, w, device=DEV)
        y = UF.grid_mask(x, d, length, st_h, st_w, use[0], use[1], mode)
        m = torch.from_numpy(R.mask(h, w, d, length, st_h, st_w, 0, use[0], use[1], mode)).to(DEV)
        assert torch.equal(y, x * m), (h, w, d, length, st_h, st_w)


def test_bad_geometry_is_refused_and_empty_batches_pass():
    import unibev_amd.functional as UF
    from unibev_amd._lib import UniBEVHipError
    x = torch.ones(2, 8, 8, device=DEV)
    with pytest.raises(UniBEVHipError):
        UF.grid_mask(x, 4, 4, 0, 0)              # l must be < d
    with pytest.raises(RuntimeError):
        UF.grid_mask(torch.ones(2, 8, 8), 4, 2, 0, 0)          # CPU tensor: no fallback
    assert UF.grid_mask(x[:0], 4, 2, 0, 0).shape == (0, 8, 8)
