import os, sys, torch, collections
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
enc = build_from_cfg(cfg, MIDDLE_ENCODERS).to(dev).train()
bs = 2
f = torch.cat([mean[:m]] * bs).float().contiguous(); zyx = coors[:m, -3:]
c = torch.cat([torch.cat((torch.full_like(zyx[:, :1], b), zyx), 1) for b in range(bs)]).contiguous()
log = collections.defaultdict(list)
def wrap(name):
    orig = getattr(UF, name)
    def fn(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record()
        key = (name, tuple(a[0].shape), tuple(a[1].shape) if name == 'spconv_wgrad' else tuple(a[1].shape))
        log[key].append((e0, e1)); return r
    setattr(UF, name, fn)
wrap('spconv_wgrad'); wrap('spconv_gather_mma')
import unibev_amd.modules.sparse_encoder as SE
for _ in range(4):
    for p in enc.parameters(): p.grad = None
    enc(f, c, bs).sum().backward()
torch.cuda.synchronize()
tot = collections.Counter()
for k, ev in sorted(log.items()):
    ts = [a.elapsed_time(b) * 1e3 for a, b in ev[len(ev) // 2:]]
    per_pass = sum(ts) / 2
    tot[k[0]] += per_pass
    print(k, 'calls/pass', len(ev) // 4, 'us/call', round(sum(ts) / len(ts), 1), 'us/pass', round(per_pass, 1))
print(dict(tot))
