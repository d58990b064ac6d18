"""The C-ABI library loads on a machine without a GPU and exports exactly the symbols
``include/unibev_hip.h`` declares; the ctypes table mirrors the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'unibev_hip.h')


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(ubv_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ('ubv_ms_deform_attn_forward', 'ubv_ms_deform_attn_backward', 'ubv_bev_lift_forward',
                 'ubv_bev_lift_backward', 'ubv_point_sampling', 'ubv_hard_voxelize',
                 'ubv_dynamic_voxelize', 'ubv_voxel_mean', 'ubv_sparse_to_dense',
                 'ubv_bev_fuse_forward', 'ubv_flatten_embed_forward'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from unibev_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(handle, s), f'{s} declared in the header but not exported'


def test_ctypes_table_matches_header():
    from unibev_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    h = _lib.lib()
    assert h.ubv_version() >= 100
    assert h.ubv_arch() == b'gfx950'
    # argument counts: parse each prototype of the header
    txt = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    for name, (res, args) in _lib.SIGNATURES.items():
        m = re.search(r'\b' + name + r'\s*\(([^;]*?)\)\s*;', txt, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        assert n == len(args), (name, n, len(args))


def test_no_compute_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from unibev_amd.functional import ms_deform_attn, hard_voxelize
    with pytest.raises(RuntimeError):
        ms_deform_attn(torch.zeros(1, 4, 8, 32), torch.tensor([[2, 2]]), torch.tensor([0]),
                       torch.zeros(1, 3, 8, 1, 4, 2), torch.zeros(1, 3, 8, 1, 4))
    with pytest.raises(RuntimeError):
        hard_voxelize(torch.zeros(10, 5), [1, 1, 1], [0, 0, 0, 4, 4, 2], 3, 10)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'unibev_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle/' not in src or f.endswith('.md'), f
