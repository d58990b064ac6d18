import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF, synthetic as syn
dev = torch.device('cuda', 0)
pts = torch.from_numpy(syn.lidar_points(30000, seed=0)).to(dev)
clouds = [torch.from_numpy(syn.lidar_points(30000, seed=s)).to(dev) for s in range(2)]
def front():
    voxels, coors, num, vnum = UF.hard_voxelize(pts, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
    return UF.voxel_mean(voxels, num, vnum)
def front_b():
    v, c, n, m_ = UF.hard_voxelize_batch(clouds, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
    return UF.voxel_mean(v.view(-1, 10, 5), n.view(-1))
for fn in (front, front_b):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(fn.__name__, 'host issue %.1f us/call, with drain %.1f us/call' % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
