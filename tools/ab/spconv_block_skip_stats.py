"""How much of the gather + MFMA work of the sparse convolutions multiplies rows without a neighbour: for every neighbour
map of the shipped-size middle encoder (bs = 2, synthetic cloud), the share of (row, offset) slots that hold a neighbour,
and the share of (row block, offset) groups with at least one — for blocks of 32 and 64 consecutive rows (an MFMA row block /
a wave) and 512 (a thread block, what the kernel skips by today)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
from unibev_amd import synthetic as syn
from unibev_amd.modules import sparse_encoder as SE
dev = torch.device('cuda', 0)
pts = torch.from_numpy(syn.lidar_points(30000, seed=0)).to(dev)
voxels, coors, num, vnum = UF.hard_voxelize(pts, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
m = int(vnum.item()); mean = UF.voxel_mean(voxels, num, vnum)
from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
cfg = dict(type='SparseEncoder', in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
           order=('conv', 'norm', 'act'), encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
           encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
enc = build_from_cfg(cfg, MIDDLE_ENCODERS).to(dev).train()
bs = 2
f = torch.cat([mean[:m]] * bs).float().contiguous(); zyx = coors[:m, -3:]
c = torch.cat([torch.cat((torch.full_like(zyx[:, :1], b), zyx), 1) for b in range(bs)]).contiguous()
seen = {}
orig = SE._SparseConv.forward
def spy(ctx, feats, weight, nbr_fwd, nbr_bwd, holder=None):
    key = (nbr_fwd.data_ptr(), tuple(nbr_fwd.shape))
    if key not in seen:
        seen[key] = (nbr_fwd, nbr_bwd, weight.shape[-2], weight.shape[-1])
    return orig(ctx, feats, weight, nbr_fwd, nbr_bwd, holder)
SE._SparseConv.forward = staticmethod(spy)
enc(f, c, bs)
def stats(nbr):
    kvol, rows = nbr.shape
    has = nbr >= 0
    out = [float(has.float().mean())]
    for blk in (32, 64, 512):
        pad = (-rows) % blk
        h = torch.nn.functional.pad(has, (0, pad)).view(kvol, -1, blk).any(-1)
        out.append(float(h.float().mean()))
    return out
for (ptr, shape), (nf, nb, cin, cout) in seen.items():
    for name, nbr in (('fwd', nf), ('bwd', nb)):
        if nbr is None: continue
        s = stats(nbr)
        print(f'{name} kvol {nbr.shape[0]:2d} rows {nbr.shape[1]:7d} cin {cin:3d} cout {cout:3d}: slots {s[0]:.3f}  blocks32 {s[1]:.3f}  blocks64 {s[2]:.3f}  blocks512 {s[3]:.3f}')
