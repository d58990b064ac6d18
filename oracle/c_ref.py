"""ORACLE — test infrastructure only.  ctypes wrapper of ``oracle/_build/liboracle.so`` (plain-C
restatements in ``msda_ref.c``, ``voxelize_ref.c`` and ``dynamic_scatter_ref.c``; build with
``make -C oracle``)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_hard_voxelize.restype = ctypes.c_int
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    pts = np.ascontiguousarray(points, np.float32)
    N, F = pts.shape
    vs = np.asarray(voxel_size, np.float32)
    rg = np.asarray(coors_range, np.float32)
    voxels = np.zeros((max_voxels, max_points, F), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    m = lib().oracle_hard_voxelize(_f(pts), N, F, _f(vs), _f(rg), max_points, max_voxels,
                                   _f(voxels), _f(coors), _f(num))
    return voxels[:m], coors[:m], num[:m]


def dynamic_voxelize(points, voxel_size, coors_range):
    pts = np.ascontiguousarray(points, np.float32)
    N, F = pts.shape
    coors = np.zeros((N, 3), np.int32)
    lib().oracle_dynamic_voxelize(_f(pts), N, F, _f(np.asarray(voxel_size, np.float32)),
                                  _f(np.asarray(coors_range, np.float32)), _f(coors))
    return coors


def dynamic_scatter(feats, coors, reduce_type):
    """-> (voxel_feats [M,C], voxel_coors [M,D], point2voxel_map [N], voxel_points_count [M]);
    reduce_type 'sum' | 'mean' | 'max'."""
    f = np.ascontiguousarray(feats, np.float32)
    c = np.ascontiguousarray(coors, np.int32)
    N, C = f.shape
    D = c.shape[1]
    of = np.zeros((max(N, 1), C), np.float32)
    oc = np.zeros((max(N, 1), D), np.int32)
    mp = np.zeros((N,), np.int32)
    cnt = np.zeros((max(N, 1),), np.int32)
    lib().oracle_dynamic_scatter.restype = ctypes.c_int
    m = lib().oracle_dynamic_scatter(_f(f), _f(c), N, C, D, {'sum': 0, 'mean': 1, 'max': 2}[reduce_type],
                                     _f(of), _f(oc), _f(mp), _f(cnt))
    return of[:m], oc[:m], mp, cnt[:m]


def dynamic_scatter_backward(grad_voxel_feats, feats, voxel_feats, point2voxel_map, voxel_points_count, reduce_type):
    """Sequential restatement of [ext] mmdet3d 0.18.1 ``dynamic_point_to_voxel_backward``
    (ops/voxel/src/scatter_points_cuda.cu: ``add_reduce_traceback_grad_kernel`` for sum / mean,
    ``max_reduce_traceback_scatter_idx_kernel`` + ``max_reduce_scatter_grad_kernel`` for max): the gradient of a
    (voxel, channel) maximum goes to ONE point, the smallest point index whose feature equals the reduced value
    (``atomicMin`` over point indices).  Source not vendored in /root/reference: parity unpinned, known-answer tested."""
    g = np.asarray(grad_voxel_feats, np.float64)
    f = np.asarray(feats, np.float32)
    N, C = f.shape
    out = np.zeros((N, C), np.float64)
    if reduce_type in ('sum', 'mean'):
        for i in range(N):
            v = int(point2voxel_map[i])
            if v >= 0:
                out[i] = g[v] / (int(voxel_points_count[v]) if reduce_type == 'mean' else 1)
        return out
    arg = np.full((len(voxel_feats), C), N, np.int64)
    for i in range(N):                       # ascending point index: the first hit is the minimum
        v = int(point2voxel_map[i])
        if v < 0:
            continue
        for ch in range(C):
            if f[i, ch] == voxel_feats[v, ch] and arg[v, ch] == N:
                arg[v, ch] = i
    for v in range(len(voxel_feats)):
        for ch in range(C):
            if arg[v, ch] < N:
                out[arg[v, ch], ch] = g[v, ch]
    return out


def voxel_mean(voxels, num_points):
    v = np.ascontiguousarray(voxels, np.float32)
    n = np.ascontiguousarray(num_points, np.int32)
    M, T, F = v.shape
    mean = np.zeros((M, F), np.float32)
    lib().oracle_voxel_mean(_f(v), _f(n), M, T, F, _f(mean))
    return mean


def sparse_to_dense(feats, coors, B, D, H, W):
    f = np.ascontiguousarray(feats, np.float32)
    c = np.ascontiguousarray(coors, np.int32)
    M, C = f.shape
    dense = np.zeros((B, C, D, H, W), np.float32)
    lib().oracle_sparse_to_dense(_f(f), _f(c), M, B, C, D, H, W, _f(dense))
    return dense


def msda_forward(value, spatial_shapes, loc, aw):
    v = np.ascontiguousarray(value, np.float32)
    B, S, H, Dh = v.shape
    ss = np.ascontiguousarray(spatial_shapes, np.int64).reshape(-1, 2)
    L = ss.shape[0]
    ls = np.concatenate(([0], np.cumsum(ss[:, 0] * ss[:, 1])[:-1])).astype(np.int64)
    lo = np.ascontiguousarray(loc, np.float32)
    w = np.ascontiguousarray(aw, np.float32)
    Nq, P = lo.shape[1], lo.shape[4]
    out = np.zeros((B, Nq, H * Dh), np.float32)
    lib().oracle_msda_forward(_f(v), _f(ss), _f(ls), _f(lo), _f(w), _f(out), B, S, H, Dh, L, Nq, P)
    return out


def msda_backward(value, spatial_shapes, loc, aw, gout):
    v = np.ascontiguousarray(value, np.float32)
    B, S, H, Dh = v.shape
    ss = np.ascontiguousarray(spatial_shapes, np.int64).reshape(-1, 2)
    L = ss.shape[0]
    ls = np.concatenate(([0], np.cumsum(ss[:, 0] * ss[:, 1])[:-1])).astype(np.int64)
    lo = np.ascontiguousarray(loc, np.float32)
    w = np.ascontiguousarray(aw, np.float32)
    g = np.ascontiguousarray(gout, np.float32)
    Nq, P = lo.shape[1], lo.shape[4]
    gv = np.zeros(v.shape, np.float64)
    gl = np.zeros(lo.shape, np.float32)
    gw = np.zeros(w.shape, np.float32)
    lib().oracle_msda_backward(_f(v), _f(ss), _f(ls), _f(lo), _f(w), _f(g), _f(gv), _f(gl), _f(gw),
                               B, S, H, Dh, L, Nq, P)
    return gv, gl, gw
