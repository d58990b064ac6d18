"""Forward-only HIP graph of the two encoders on two streams, replayed: is the output the one-stream output?
   python tools/ab/fwd_repro_graph.py [replays]   (UBV_TWO_STREAMS=1 to exercise the two-stream mode)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
import test_modules_gpu as T
from _util import encoder_case, t, tq
from unibev_amd.modules import transformer as TR
DEV = 'cuda'
torch.cuda.set_stream(torch.cuda.Stream())
cfg, sd, inp, g = encoder_case('fullsize_smooth')
model = T._build(cfg).to(DEV).eval()
T._load(model, sd)
gi = [t(x, device=DEV) for x in inp['img']]
gp = [t(x, device=DEV) for x in inp['pts']]
gq = tq(inp['bev_q'], device=DEV)
bev_pos = t(inp['bev_pos'], device=DEV)
def fwd():
    with torch.no_grad():
        return model.encode(gi, gp, gq, inp['bev_h'], inp['bev_w'], bev_pos=bev_pos, img_metas=inp['metas'], return_parts=True)
TR.set_two_streams(False)
ref = [x.clone() for x in fwd()]
TR.set_two_streams(os.environ.get('UBV_TWO_STREAMS', '1') != '0')
for _ in range(3):
    fwd()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=torch.cuda.current_stream(), capture_error_mode='thread_local'):
    out = fwd()
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for i in range(n):
    gr.replay()
    torch.cuda.synchronize()
    d = [float((a - b).norm() / b.norm()) for a, b in zip(out, ref)]
    bad += any(v > 1e-6 for v in d)
    print(f'replay {i}: fused {d[0]:.1e} img {d[1]:.1e} pts {d[2]:.1e}')
print(f'{bad} of {n} replays differ from the one-stream output')
