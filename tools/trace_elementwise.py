"""Traces the call sites of the large element-wise framework ops of one f32 training step (TorchDispatchMode)."""
import os, sys, torch, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
dev = torch.device('cuda', 0)
from unibev_amd.modules import transformer as TR
TR.set_two_streams(False)
head, _ = B.build_head('LC_cnw', dev)
img, pts, metas = B.synth_inputs('LC_cnw', 2, torch.float32, dev, 0)
params = [p for p in head.parameters() if p.requires_grad]
cot = torch.randn(200 * 200, 2, 256, device=dev) / 200.0
head.transformer.forced_flags = (1, 1)
def step():
    for p in params: p.grad = None
    out = head.forward_bev(img, pts, metas)
    out.backward(cot.to(out.dtype).view_as(out) if cot.numel() == out.numel() else cot)   # the bench's step: cotangent fed directly
for _ in range(2): step()
log = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat_args = [t for a in args for t in (a if isinstance(a, (list, tuple)) else [a])]       # (cat / stack take lists)
        mn = int(os.environ.get('UBV_TRACE_MIN', 5_000_000))
        big = [a for a in flat_args if isinstance(a, torch.Tensor) and a.numel() >= mn]
        if not big and 'cat' in name and isinstance(out, torch.Tensor) and out.numel() >= mn // 8:
            big = [out]
        gemm = any(k in name for k in ('aten.mm', 'aten.addmm', 'aten.bmm', 'aten.linear', 'aten.matmul'))
        if gemm:
            big = [a for a in args if isinstance(a, torch.Tensor)]
        if big and (gemm or any(k in name for k in ('add', 'clone', 'copy', 'mul', 'sum', 'cat', 'contiguous', 'div', 'to.', '_to_copy', 'zero', 'fill', 'softmax', 'index', 'stack', 'sub', 'neg', 'where', 'slice', 'select', 'new_', 'expand', 'permute_copy', 'masked'))):
            st = traceback.extract_stack()
            site = [f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in st if 'unibev_amd' in f.filename][-2:]
            log[(name, tuple(big[0].shape), tuple(big[0].stride()), ' <- '.join(reversed(site)) or 'autograd engine')] += 1
        return out
with Mode():
    step()
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
