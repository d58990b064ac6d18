"""Sparse middle encoder, shipped-config size, bs = 2: forward and forward + backward wall time (HIP events, rulebooks cached)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
from unibev_amd import synthetic as syn
dev = torch.device('cuda', 0)
pts = torch.from_numpy(syn.lidar_points(30000, seed=0)).to(dev)
voxels, coors, num, vnum = UF.hard_voxelize(pts, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
m = int(vnum.item()); mean = UF.voxel_mean(voxels, num, vnum)
from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
cfg = dict(type='SparseEncoder', in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
           order=('conv', 'norm', 'act'), encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
           encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
torch.manual_seed(0)
enc = build_from_cfg(cfg, MIDDLE_ENCODERS).to(dev).train()
enc.keep_rulebooks = os.environ.get('UBV_KEEP_RULEBOOKS', '1') != '0'
bs = 2
f = torch.cat([mean[:m]] * bs).float().contiguous(); zyx = coors[:m, -3:]
c = torch.cat([torch.cat((torch.full_like(zyx[:, :1], b), zyx), 1) for b in range(bs)]).contiguous()
def fb():
    for p in enc.parameters(): p.grad = None
    enc(f, c, bs).sum().backward()
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(fb)
g = torch.cat([p.grad.flatten() for p in enc.parameters() if p.grad is not None])
print(os.environ.get('TAG', ''), f'forward + backward {t:.2f} ms   grad checksum {float(g.double().abs().sum()):.6e}')
