"""LiDAR front end on the GPU vs the sequential C oracle: voxel indices bit-exact."""
import numpy as np
import pytest
import torch

from _util import t
from oracle import c_ref
from unibev_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda'
VS = [0.075, 0.075, 0.2]
RG = [-54, -54, -5, 54, 54, 3]


def run_gpu(points, T, M, vs=VS, rg=RG):
    from unibev_amd.functional import hard_voxelize
    voxels, coors, num, vnum = hard_voxelize(t(points, device=DEV), vs, rg, T, M)
    m = int(vnum.item())
    return voxels[:m].cpu().numpy(), coors[:m].cpu().numpy(), num[:m].cpu().numpy(), \
        (voxels, coors, num, vnum)


def check_equal(points, T, M, vs=VS, rg=RG):
    v_ref, c_ref_, n_ref = c_ref.hard_voxelize(points, vs, rg, T, M)
    v, c, n, raw = run_gpu(points, T, M, vs, rg)
    assert c.dtype == np.int32 and n.dtype == np.int32
    np.testing.assert_array_equal(c, c_ref_)
    np.testing.assert_array_equal(n, n_ref)
    np.testing.assert_array_equal(v, v_ref)
    return raw


def test_synthetic_30k_cloud_bit_exact():
    raw = check_equal(syn.lidar_points(30000, seed=0), 10, 90000)
    voxels, coors, num, vnum = raw
    # rows past voxel_num stay zero
    m = int(vnum.item())
    assert torch.all(voxels[m:] == 0) and torch.all(num[m:] == 0)


def test_dense_cloud_many_points_per_voxel_and_voxel_cap():
    rs = np.random.RandomState(3)
    pts = np.concatenate([rs.uniform(-2, 2, (60000, 2)), rs.uniform(-1.5, -0.5, (60000, 1)),
                          rs.uniform(0, 1, (60000, 2))], 1).astype(np.float32)
    check_equal(pts, 10, 90000)
    check_equal(pts, 3, 500)          # voxel budget exhausted: later first-appearances dropped
    check_equal(pts, 1, 90000)


def test_known_answer_cloud():
    """Hand-built cloud: duplicates, a point exactly on the max edge (dropped), on the min edge
    (kept), NaN, outside z."""
    vs, rg = [1.0, 1.0, 1.0], [0, 0, 0, 4, 4, 2]
    pts = np.array([
        [0.5, 0.5, 0.5, 1, 0], [3.5, 0.5, 0.5, 2, 0], [0.6, 0.4, 0.9, 3, 0], [4.0, 1.0, 1.0, 4, 0],
        [0.0, 0.0, 0.0, 5, 0], [1.5, 2.5, 1.5, 6, 0], [np.nan, 1.0, 1.0, 7, 0], [1.0, 1.0, 2.0, 8, 0],
        [1.5, 2.5, 1.2, 9, 0], [0.9, 0.9, 0.1, 10, 0], [-0.0001, 1.0, 1.0, 11, 0], [3.999, 3.999, 1.999, 12, 0],
    ], np.float32)
    v, c, n, _ = run_gpu(pts, 3, 10, vs, rg)
    np.testing.assert_array_equal(c, [[0, 0, 0], [0, 0, 3], [1, 2, 1], [1, 3, 3]])
    np.testing.assert_array_equal(n, [3, 1, 2, 1])       # voxel 0 saw 4 points, keeps the first 3
    np.testing.assert_array_equal(v[0, :, 3], [1, 3, 5])
    np.testing.assert_array_equal(v[2, :2, 3], [6, 9])
    assert np.all(v[1, 1:] == 0)
    check_equal(pts, 3, 10, vs, rg)


def test_empty_and_all_outside():
    v, c, n, (voxels, coors, num, vnum) = run_gpu(np.zeros((0, 5), np.float32), 10, 100)
    assert v.shape[0] == 0 and int(vnum.item()) == 0
    pts = np.full((100, 5), 1e4, np.float32)
    v, c, n, _ = run_gpu(pts, 10, 100)
    assert v.shape[0] == 0


def test_dynamic_voxelize_mean_vfe_and_dense_scatter():
    from unibev_amd.functional import dynamic_voxelize, voxel_mean, sparse_to_dense
    pts = syn.lidar_points(20000, seed=2)
    np.testing.assert_array_equal(dynamic_voxelize(t(pts, device=DEV), VS, RG).cpu().numpy(),
                                  c_ref.dynamic_voxelize(pts, VS, RG))
    v_ref, c_ref_, n_ref = c_ref.hard_voxelize(pts, VS, RG, 10, 90000)
    mean = voxel_mean(t(v_ref, device=DEV), t(n_ref, device=DEV))
    np.testing.assert_allclose(mean.cpu().numpy(), c_ref.voxel_mean(v_ref, n_ref), rtol=1e-6, atol=1e-6)
    rs = np.random.RandomState(0)
    M, C = 5000, 16
    cells = rs.choice(2 * 2 * 180 * 180, M, replace=False)
    coors = np.stack([cells // (2 * 180 * 180), (cells // (180 * 180)) % 2, (cells // 180) % 180,
                      cells % 180], 1).astype(np.int32)
    feats = rs.standard_normal((M, C)).astype(np.float32)
    dense = sparse_to_dense(t(feats, device=DEV), t(coors, device=DEV), 2, (2, 180, 180))
    np.testing.assert_array_equal(dense.cpu().numpy(), c_ref.sparse_to_dense(feats, coors, 2, 2, 180, 180))


def test_voxelization_module_and_batch_helper():
    from unibev_amd.modules.voxel import Voxelization, HardSimpleVFE, voxelize_batch
    layer = Voxelization(VS, RG, 10, (90000, 120000)).eval()
    assert layer.grid_size.tolist() == [1440, 1440, 40]
    clouds = [syn.lidar_points(5000, seed=s) for s in (4, 5)]
    voxels, num, coors = voxelize_batch(layer, [t(c, device=DEV) for c in clouds])
    ref = [c_ref.hard_voxelize(c, VS, RG, 10, 120000) for c in clouds]
    np.testing.assert_array_equal(voxels.cpu().numpy(), np.concatenate([r[0] for r in ref]))
    np.testing.assert_array_equal(coors[:, 1:].cpu().numpy(), np.concatenate([r[1] for r in ref]))
    np.testing.assert_array_equal(coors[:, 0].cpu().numpy(),
                                  np.concatenate([np.full(len(r[1]), i) for i, r in enumerate(ref)]))
    feats = HardSimpleVFE(5)(voxels, num, coors)
    np.testing.assert_allclose(feats.cpu().numpy(),
                               np.concatenate([c_ref.voxel_mean(r[0], r[2]) for r in ref]),
                               rtol=1e-6, atol=1e-6)


def test_full_sweep_cloud_idempotent_property():
    """~300k points (10-sweep size): voxelizing the voxel contents again reproduces the voxels
    (idempotence), and every stored point lies in its voxel's cell."""
    pts = syn.lidar_points(300000, seed=7)
    v, c, n, _ = run_gpu(pts, 10, 120000)
    flat = np.concatenate([v[i, :n[i]] for i in range(0, len(v), 97)])
    cc = c_ref.dynamic_voxelize(flat, VS, RG)
    owner = np.concatenate([np.repeat(c[i][None], n[i], 0) for i in range(0, len(v), 97)])
    np.testing.assert_array_equal(cc, owner)
    check_equal(pts, 10, 120000)


@pytest.mark.parametrize('reduce_type', ['sum', 'mean', 'max'])
@pytest.mark.parametrize('batched', [False, True])
def test_dynamic_scatter_bit_exact_vs_oracle(reduce_type, batched):
    """ubv_dynamic_point_to_voxel_forward on the dynamic voxelization of the synthetic cloud(s):
    voxel coordinates, point->voxel map and counts bit-exact, features bit-exact too (both sides
    reduce a voxel's points in input order)."""
    from unibev_amd.functional import dynamic_scatter, dynamic_voxelize
    clouds = [syn.lidar_points(30000, seed=s) for s in ((0, 1) if batched else (0,))]
    coors, feats = [], []
    for b, cloud in enumerate(clouds):
        c = dynamic_voxelize(t(cloud, device=DEV), syn.VOXEL_SIZE, syn.PC_RANGE).cpu().numpy()
        np.testing.assert_array_equal(c, c_ref.dynamic_voxelize(cloud, syn.VOXEL_SIZE, syn.PC_RANGE))
        if batched:
            c = np.concatenate([np.where((c < 0).any(1, keepdims=True), -1, b).astype(np.int32), c], 1)
        coors.append(c)
        feats.append(cloud)
    coors, feats = np.concatenate(coors), np.concatenate(feats)
    vf, vc, mp, cnt, vnum = dynamic_scatter(t(feats, device=DEV), t(coors, device=DEV), reduce_type)
    m, nvalid = (int(v) for v in vnum.cpu())
    ef, ec, emp, ecnt = c_ref.dynamic_scatter(feats, coors, reduce_type)
    assert m == len(ec) and nvalid == int((emp >= 0).sum()) and m > 10000
    np.testing.assert_array_equal(vc[:m].cpu().numpy(), ec)
    np.testing.assert_array_equal(mp.cpu().numpy(), emp)
    np.testing.assert_array_equal(cnt[:m].cpu().numpy(), ecnt)
    np.testing.assert_array_equal(vf[:m].cpu().numpy(), ef)


def test_dynamic_scatter_edge_cases_and_module():
    from unibev_amd.functional import dynamic_scatter
    from unibev_amd.modules.voxel import DynamicScatter
    # nothing valid, a single voxel, an empty input
    f = torch.arange(12, dtype=torch.float32, device=DEV).view(4, 3)
    vf, vc, mp, cnt, vnum = dynamic_scatter(f, torch.full((4, 3), -1, dtype=torch.int32, device=DEV), 'max')
    assert vnum.tolist() == [0, 0] and (mp == -1).all()
    vf, vc, mp, cnt, vnum = dynamic_scatter(f, torch.tensor([[2, 3, 4]] * 4, dtype=torch.int32, device=DEV), 'max')
    assert vnum.tolist() == [1, 4] and vc[0].tolist() == [2, 3, 4] and vf[0].tolist() == [9.0, 10.0, 11.0]
    assert cnt[0].item() == 4 and (mp == 0).all()
    vf, vc, mp, cnt, vnum = dynamic_scatter(f[:0], torch.zeros(0, 3, dtype=torch.int32, device=DEV), 'sum')
    assert vnum.tolist() == [0, 0]
    with pytest.raises(ValueError):
        dynamic_scatter(f, torch.zeros(4, 3, dtype=torch.int32, device=DEV), 'median')
    # module: exact shapes of the published op, gradient of the mean goes back divided by the count
    coors = torch.tensor([[1, 0, 2], [0, 5, 5], [1, 0, 2], [-1, 3, 3], [0, 5, 5], [0, 0, 9]],
                         dtype=torch.int32, device=DEV)
    feats = torch.arange(12, dtype=torch.float32, device=DEV).view(6, 2).requires_grad_()
    layer = DynamicScatter([0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], True)
    vfeat, vcoor = layer(feats, coors)
    assert vcoor.tolist() == [[0, 0, 9], [0, 5, 5], [1, 0, 2]]
    assert vfeat.tolist() == [[10.0, 11.0], [5.0, 6.0], [2.0, 3.0]]
    vfeat.sum().backward()
    assert feats.grad.tolist() == [[0.5, 0.5], [0.5, 0.5], [0.5, 0.5], [0.0, 0.0], [0.5, 0.5], [1.0, 1.0]]


@pytest.mark.parametrize('reduce_type', ['sum', 'mean', 'max'])
def test_dynamic_scatter_backward_with_ties_vs_sequential_oracle(reduce_type):
    """LiDAR channels tie constantly (quantised intensity, all-zero time stamps): the max gradient must reach exactly
    one point per (voxel, channel) — the smallest index attaining the maximum — as the published op traces it back."""
    from unibev_amd.modules.voxel import dynamic_scatter as ds
    rs = np.random.RandomState(5)
    n = 4000
    coors = rs.randint(0, 6, (n, 3)).astype(np.int32)
    coors[rs.rand(n) < 0.05] = -1
    feats = np.stack([rs.randint(0, 4, n), np.zeros(n), rs.standard_normal(n), rs.randint(0, 2, n)], 1).astype(np.float32)
    fg = t(feats, device=DEV).requires_grad_()
    vf, vc = ds(fg, t(coors, device=DEV), reduce_type)
    cot = rs.standard_normal(tuple(vf.shape)).astype(np.float32)
    (vf * t(cot, device=DEV)).sum().backward()
    ef, ec, emp, ecnt = c_ref.dynamic_scatter(feats, coors, reduce_type)
    np.testing.assert_array_equal(vf.detach().cpu().numpy(), ef)
    want = c_ref.dynamic_scatter_backward(cot, feats, ef, emp, ecnt, reduce_type)
    np.testing.assert_allclose(fg.grad.cpu().numpy(), want, rtol=1e-6, atol=0)
    if reduce_type == 'max':                 # one receiver per (voxel, channel)
        assert int((fg.grad != 0).sum()) <= ef.size and float(fg.grad[:, 1].abs().sum()) > 0
    # no valid point at all: zero gradient, no error
    fz = t(feats[:7], device=DEV).requires_grad_()
    vz, _ = ds(fz, torch.full((7, 3), -1, dtype=torch.int32, device=DEV), reduce_type)
    assert vz.shape[0] == 0
    vz.sum().backward()
    assert fz.grad is not None and float(fz.grad.abs().sum()) == 0


def test_hard_voxelize_batch_is_bit_identical_to_per_sample_calls():
    """One launch chain for a list of clouds (``ubv_hard_voxelize_batch``): every sample equals its own
    ``ubv_hard_voxelize`` call and the C oracle bit for bit — clouds of different sizes, an empty one, a voxel cap that
    bites; ``voxelize_cat`` gives the reference's concatenated (voxels, num_points, coors [b, z, y, x])."""
    from unibev_amd.functional import hard_voxelize, hard_voxelize_batch
    from unibev_amd.modules.voxel import Voxelization, voxelize_cat
    clouds = [syn.lidar_points(30000, seed=3), syn.lidar_points(7000, seed=4), np.zeros((0, 5), np.float32),
              syn.lidar_points(19000, seed=5)]
    dev = [t(c, device=DEV) for c in clouds]
    for cap in (90000, 5000):
        v, c, n, m = hard_voxelize_batch(dev, syn.VOXEL_SIZE, syn.PC_RANGE, 10, cap)
        assert v.shape == (4, cap, 10, 5) and m.shape == (4,)
        for b, cloud in enumerate(clouds):
            ev, ec, en = c_ref.hard_voxelize(cloud, syn.VOXEL_SIZE, syn.PC_RANGE, 10, cap)
            k = int(m[b])
            assert k == len(ec)
            np.testing.assert_array_equal(c[b, :k].cpu().numpy(), ec)
            np.testing.assert_array_equal(n[b, :k].cpu().numpy(), en)
            np.testing.assert_array_equal(v[b, :k].cpu().numpy(), ev)
            if len(cloud):
                sv, sc, sn, sm = hard_voxelize(dev[b], syn.VOXEL_SIZE, syn.PC_RANGE, 10, cap)
                assert int(sm) == k and torch.equal(sv[:k], v[b, :k]) and torch.equal(sc[:k], c[b, :k])
    layer = Voxelization(syn.VOXEL_SIZE, syn.PC_RANGE, 10, (90000, 120000)).eval()
    voxels, num, coors = voxelize_cat(layer, dev)
    per = [layer(p) for p in dev]
    assert torch.equal(voxels, torch.cat([p[0] for p in per])) and torch.equal(num, torch.cat([p[2] for p in per]))
    assert torch.equal(coors[:, 1:], torch.cat([p[1] for p in per]))
    assert coors[:, 0].tolist() == sum(([b] * len(p[1]) for b, p in enumerate(per)), [])


def test_voxelization_chain_with_the_vfe_mean_on_the_way():
    """``ubv_hard_voxelize_batch_vfe``: the HardSimpleVFE means written by the chain's gather launch equal
    ``ubv_voxel_mean`` of the voxel slab bit for bit (dense voxels with up to T points, a voxel cap that bites, an empty
    cloud), rows past the count are zero, and ``extract_pts_feat`` takes them from there."""
    from unibev_amd.functional import hard_voxelize_batch, voxel_mean
    from unibev_amd.modules.voxel import HardSimpleVFE, Voxelization, voxelize_cat
    rs = np.random.RandomState(5)
    dense = np.concatenate([rs.uniform(-2, 2, (40000, 2)), rs.uniform(-1.5, -0.5, (40000, 1)),
                            rs.uniform(0, 1, (40000, 2))], 1).astype(np.float32)
    clouds = [syn.lidar_points(30000, seed=6), dense, np.zeros((0, 5), np.float32)]
    dev = [t(c, device=DEV) for c in clouds]
    for cap in (90000, 700):
        v, c, n, m, mean = hard_voxelize_batch(dev, syn.VOXEL_SIZE, syn.PC_RANGE, 10, cap, with_mean=True)
        v0, c0, n0, m0 = hard_voxelize_batch(dev, syn.VOXEL_SIZE, syn.PC_RANGE, 10, cap)
        assert torch.equal(v, v0) and torch.equal(n, n0) and torch.equal(m, m0) and torch.equal(c, c0)
        for b in range(len(clouds)):
            ref = voxel_mean(v[b], n[b], m[b:b + 1])
            assert torch.equal(mean[b], ref)
            assert torch.all(mean[b, int(m[b]):] == 0)
    layer = Voxelization(syn.VOXEL_SIZE, syn.PC_RANGE, 10, (90000, 120000)).eval()
    voxels, num, coors, mean = voxelize_cat(layer, dev, with_mean=True)
    assert torch.equal(mean[:, :4].contiguous(), HardSimpleVFE(4)(voxels, num, coors))
