"""One sparse convolution shape of the middle encoder (128 -> 128 channels, 3 x 3 x 3 submanifold, bs = 2): us per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
from unibev_amd import synthetic as syn
dev = torch.device('cuda', 0)
pts = torch.from_numpy(syn.lidar_points(30000, seed=0)).to(dev)
voxels, coors, num, vnum = UF.hard_voxelize(pts, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
m = int(vnum.item())
bs = 2
zyx = coors[:m, -3:]
c = torch.cat([torch.cat((torch.full_like(zyx[:, :1], b), zyx), 1) for b in range(bs)]).contiguous().int()
dims = (41, 1440, 1440)
for (k, s, p) in (((3, 3, 3), (2, 2, 2), (1, 1, 1)),) * 3:
    c, dims, _, _ = UF.spconv_strided_maps(c, bs, dims, k, s, p)
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nbr = UF.spconv_subm_map(c, bs, dims, (3, 3, 3))
rows = c.shape[0]
torch.manual_seed(0)
f = torch.randn(rows, ch, device=dev)
w = torch.randn(27, ch, ch, device=dev) * 0.05
hi, lo = UF.spconv_operand(w)
if len(sys.argv) > 2 and sys.argv[2] == 'wgrad':
    g = torch.randn(rows, ch, device=dev)
    pairs = UF.spconv_pairs(nbr)
    for _ in range(3): out = UF.spconv_wgrad(g, f, nbr, pairs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): out = UF.spconv_wgrad(g, f, nbr, pairs)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get('TAG', ''), f'wgrad rows {rows} pairs {int(pairs[2].sum())}  {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {float(out.double().abs().sum()):.6e}')
    sys.exit(0)
for _ in range(3): out = UF.spconv_gather_mma(f, nbr, hi, lo, ch)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): out = UF.spconv_gather_mma(f, nbr, hi, lo, ch)
e1.record(); torch.cuda.synchronize()
dens = float((nbr >= 0).float().mean())
print(os.environ.get('TAG', ''), f'rows {rows} dims {dims} density {dens:.3f}  {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {float(out.double().abs().sum()):.6e}')
