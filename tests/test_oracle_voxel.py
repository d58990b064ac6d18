"""The C voxelization oracle against hand-built known-answer clouds and defining properties
(the reference holds no vectors for this op: SURVEY.md section 8(c))."""
import numpy as np

from oracle import c_ref
from unibev_amd import synthetic as syn

VS = [0.075, 0.075, 0.2]
RG = [-54, -54, -5, 54, 54, 3]


def test_known_answer_cloud():
    vs, rg = [1.0, 1.0, 1.0], [0, 0, 0, 4, 4, 2]
    pts = np.array([
        [0.5, 0.5, 0.5, 1, 0], [3.5, 0.5, 0.5, 2, 0], [0.6, 0.4, 0.9, 3, 0], [4.0, 1.0, 1.0, 4, 0],
        [0.0, 0.0, 0.0, 5, 0], [1.5, 2.5, 1.5, 6, 0], [np.nan, 1.0, 1.0, 7, 0], [1.0, 1.0, 2.0, 8, 0],
        [1.5, 2.5, 1.2, 9, 0], [0.9, 0.9, 0.1, 10, 0], [-0.0001, 1.0, 1.0, 11, 0], [3.999, 3.999, 1.999, 12, 0],
    ], np.float32)
    v, c, n = c_ref.hard_voxelize(pts, vs, rg, 3, 10)
    np.testing.assert_array_equal(c, [[0, 0, 0], [0, 0, 3], [1, 2, 1], [1, 3, 3]])
    np.testing.assert_array_equal(n, [3, 1, 2, 1])
    np.testing.assert_array_equal(v[0, :, 3], [1, 3, 5])
    np.testing.assert_array_equal(v[2, :2, 3], [6, 9])
    # voxel budget: third new voxel is skipped, but later points of existing voxels still land
    v, c, n = c_ref.hard_voxelize(pts, vs, rg, 3, 2)
    np.testing.assert_array_equal(c, [[0, 0, 0], [0, 0, 3]])
    np.testing.assert_array_equal(n, [3, 1])
    d = c_ref.dynamic_voxelize(pts, vs, rg)
    np.testing.assert_array_equal(d[3], [-1, -1, -1])
    np.testing.assert_array_equal(d[11], [1, 3, 3])


def test_defining_properties_on_the_synthetic_cloud():
    pts = syn.lidar_points(30000, seed=0)
    v, c, n = c_ref.hard_voxelize(pts, VS, RG, 10, 90000)
    d = c_ref.dynamic_voxelize(pts, VS, RG)
    # coordinate formula in float32: floor((p - min) / size)
    exp = np.floor((pts[:, :3] - np.float32(RG[:3])).astype(np.float32)
                   / np.float32(VS)).astype(np.int64)[:, ::-1]
    inside = (exp >= 0).all(1) & (exp < np.array([40, 1440, 1440])).all(1)
    np.testing.assert_array_equal(d[inside], exp[inside])
    assert (d[~inside] == -1).all() and (~inside).sum() >= 600
    # order of first appearance
    _, first = np.unique(d[inside], axis=0, return_index=True)
    np.testing.assert_array_equal(c, d[inside][np.sort(first)])
    # per-voxel counts capped at T, points kept in input order
    keys = {tuple(k): i for i, k in enumerate(c)}
    cnt = np.zeros(len(c), int)
    for k in d[inside]:
        cnt[keys[tuple(k)]] += 1
    np.testing.assert_array_equal(n, np.minimum(cnt, 10))
    assert n.max() <= 10 and n.min() >= 1
    m = c_ref.voxel_mean(v, n)
    np.testing.assert_allclose(m, v.sum(1) / n[:, None], rtol=1e-6)


def _np_scatter(feats, coors, reduce_type):
    """numpy statement of the published op: np.unique(axis=0) is torch.unique_dim's sorted unique."""
    ok = (coors >= 0).all(1)
    uniq, inv = np.unique(coors[ok], axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    mp = np.full(len(coors), -1, np.int32)
    mp[ok] = inv
    cnt = np.bincount(inv, minlength=len(uniq)).astype(np.int32)
    out = np.zeros((len(uniq), feats.shape[1]), np.float64)
    if reduce_type == 'max':
        out[:] = -np.inf
        np.maximum.at(out, inv, feats[ok].astype(np.float64))
    else:
        np.add.at(out, inv, feats[ok].astype(np.float64))
        if reduce_type == 'mean':
            out /= cnt[:, None]
    return out, uniq.astype(np.int32), mp, cnt


def test_dynamic_scatter_oracle_vs_numpy_unique():
    """The C restatement of dynamic_point_to_voxel_forward against numpy's unique (what the
    published op builds on): coordinates, map and counts identical, features to f32 round-off."""
    rs = np.random.RandomState(4)
    for D, N in ((3, 5000), (4, 3000), (3, 1), (2, 40)):
        coors = rs.randint(0, 9, size=(N, D)).astype(np.int32)
        coors[rs.random_sample(N) < 0.1, rs.randint(0, D)] = -1          # dropped points
        feats = rs.standard_normal((N, 5)).astype(np.float32)
        for red in ('sum', 'mean', 'max'):
            vf, vc, mp, cnt = c_ref.dynamic_scatter(feats, coors, red)
            ef, ec, emp, ecnt = _np_scatter(feats, coors, red)
            np.testing.assert_array_equal(vc, ec)
            np.testing.assert_array_equal(mp, emp)
            np.testing.assert_array_equal(cnt, ecnt)
            np.testing.assert_allclose(vf, ef, rtol=1e-5, atol=1e-5)
    # known answer
    coors = np.array([[1, 0, 2], [0, 5, 5], [1, 0, 2], [-1, 3, 3], [0, 5, 5], [0, 0, 9]], np.int32)
    feats = np.arange(12, dtype=np.float32).reshape(6, 2)
    vf, vc, mp, cnt = c_ref.dynamic_scatter(feats, coors, 'mean')
    np.testing.assert_array_equal(vc, [[0, 0, 9], [0, 5, 5], [1, 0, 2]])
    np.testing.assert_array_equal(mp, [2, 1, 2, -1, 1, 0])
    np.testing.assert_array_equal(cnt, [1, 2, 2])
    np.testing.assert_array_equal(vf, [[10, 11], [5, 6], [2, 3]])


def test_dynamic_scatter_backward_oracle_known_answers():
    """The max gradient goes to the FIRST point attaining the maximum (published op: atomicMin over point indices),
    sum copies, mean divides by the count; invalid points get nothing."""
    feats = np.array([[1, 0], [3, 0], [3, 0], [2, 0], [9, 9]], np.float32)
    mp = np.array([0, 0, 0, 0, -1])
    vf, cnt, g = np.array([[3, 0]], np.float32), np.array([4]), np.array([[5.0, 7.0]])
    np.testing.assert_array_equal(c_ref.dynamic_scatter_backward(g, feats, vf, mp, cnt, 'max'),
                                  [[0, 7], [5, 0], [0, 0], [0, 0], [0, 0]])
    np.testing.assert_array_equal(c_ref.dynamic_scatter_backward(g, feats, vf, mp, cnt, 'sum'),
                                  [[5, 7]] * 4 + [[0, 0]])
    np.testing.assert_array_equal(c_ref.dynamic_scatter_backward(g, feats, vf, mp, cnt, 'mean'),
                                  [[1.25, 1.75]] * 4 + [[0, 0]])
