#!/usr/bin/env python3
"""Micro-benchmark of the LiDAR front end at the BASELINE shapes: ~30 k points per sweep, voxel size
(0.075, 0.075, 0.2) over the 108 m x 108 m x 8 m range, max 10 points per voxel, 120 000 voxels
(configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:186-193), then the per-voxel mean and the
dense scatter of a 128-channel sparse tensor onto the 180x180x2 grid (the tail of SparseEncoder).
Times are per call, HIP events around torch-stream launches, median of N."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unibev_amd import functional as UF      # noqa: E402
from unibev_amd import synthetic as syn      # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(1e3 * s.elapsed_time(e))
    return float(np.median(ts))


def main():
    dev = 'cuda'
    pts = torch.from_numpy(syn.lidar_points(30000, seed=0)).to(dev)
    vs, rng = [0.075, 0.075, 0.2], list(syn.PC_RANGE)
    out = {}
    out['hard_voxelize (30k pts, T=10, 120k cap)'] = timed(lambda: UF.hard_voxelize(pts, vs, rng, 10, 120000))
    out['dynamic_voxelize (30k pts)'] = timed(lambda: UF.dynamic_voxelize(pts, vs, rng))
    voxels, coors, num, vnum = UF.hard_voxelize(pts, vs, rng, 10, 120000)
    out['voxel_mean (HardSimpleVFE)'] = timed(lambda: UF.voxel_mean(voxels, num, vnum))
    M = int(vnum.item())
    feats = torch.randn(M, 128, device=dev)
    c4 = torch.zeros(M, 4, dtype=torch.int32, device=dev)
    c4[:, 1] = coors[:M, 0] % 2
    c4[:, 2] = coors[:M, 1] % 180
    c4[:, 3] = coors[:M, 2] % 180
    out['sparse_to_dense (M=%d x 128 -> 1x128x2x180x180)' % M] = timed(
        lambda: UF.sparse_to_dense(feats, c4, 1, (2, 180, 180)))
    for k, v in out.items():
        print(f'{k:60s} {v:8.1f} us')
    print(f'voxels: {M}')


if __name__ == '__main__':
    main()
