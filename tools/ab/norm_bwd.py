"""add + dropout + LayerNorm at 80 000 x 256 f32: forward and backward time (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unibev_amd import functional as UF
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
R, C = 80000, 256
x = torch.randn(R, C, device='cuda', requires_grad=True); idn = torch.randn(R, C, device='cuda', requires_grad=True)
g = torch.ones(C, device='cuda', requires_grad=True); b = torch.zeros(C, device='cuda', requires_grad=True)
go = torch.randn(R, C, device='cuda')
y = UF.add_dropout_layernorm(x, idn, g, b, p=0.1, training=True)
tf = timeit(lambda: UF.add_dropout_layernorm(x, idn, g, b, p=0.1, training=True))
def bwd():
    y.backward(go, retain_graph=True)
tb = timeit(bwd)
print(os.environ.get('TAG', ''), f'forward {tf:.1f} us   backward (incl. autograd accumulation) {tb:.1f} us')
