import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from torch.profiler import profile, ProfilerActivity
class A: pass
args = A(); args.workload='LC_cnw'; args.bs=2; args.fp32_stream=False; args.eval_mode=False
dev = torch.device('cuda', 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
from unibev_amd.modules import transformer as TR
TR.set_two_streams(False)
head, _ = B.build_head('LC_cnw', dev)
dt = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dtype = B.DTYPES[dt]
img, pts, metas = B.synth_inputs('LC_cnw', 2, dtype, dev, 0)
params = [p for p in head.parameters() if p.requires_grad]
cot = torch.randn(200 * 200, 2, 256, device=dev) / 200.0
def step():
    for p in params: p.grad = None
    with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
        out = head.forward_bev(img, pts, metas)
    (out.float() * cot).sum().backward()
head.transformer.forced_flags = (1, 1)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::sum', 'aten::cat', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::to', 'aten::_to_copy'):
        dur = e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total
        if dur and dur > 8:
            st = [s for s in (e.stack or []) if 'unibev_amd' in s or 'autograd' in s][:3]
            rows.append((e.name, str(e.input_shapes)[:70], round(dur, 1), ' <- '.join(s.split('/')[-1][:60] for s in st)))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for n, sh, d, st in rows:
    agg[(n, sh, st)][0] += 1; agg[(n, sh, st)][1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{v[1]:8.1f} us  x{v[0]:3d}  {k[0]:16s} {k[1]:70s} {k[2]}')
