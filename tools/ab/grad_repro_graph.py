"""Reproducibility of the gradients under HIP-graph replay with two streams (full-size smooth fixture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
import test_modules_gpu as T
from _util import encoder_case, t, tq
from unibev_amd import synthetic as syn
from unibev_amd.graph_step import GraphedStep
from unibev_amd.modules import transformer as TR
DEV = 'cuda'
torch.cuda.set_stream(torch.cuda.Stream())
cfg, sd, inp, g = encoder_case('fullsize_smooth')
nq, bs = inp['bev_h'] * inp['bev_w'], inp['bs']
cot = t(syn.seeded_array('cot:fullsize_smooth', (nq, bs, cfg['embed_dims']), 5) / nq ** 0.5, device=DEV)
model = T._build(cfg).to(DEV).eval()
T._load(model, sd)
model.forced_flags = (1, 1)
gi = [t(x, device=DEV).requires_grad_() for x in inp['img']]
gp = [t(x, device=DEV).requires_grad_() for x in inp['pts']]
gq = tq(inp['bev_q'], device=DEV, grad=True)
bev_pos = t(inp['bev_pos'], device=DEV)
named = [(k, p) for k, p in model.named_parameters() if not k.startswith('decoder') and not k.startswith('reference_points')]
params = [p for _, p in named]
fwd = lambda: model.encode(gi, gp, gq, inp['bev_h'], inp['bev_w'], bev_pos=bev_pos, img_metas=inp['metas'])
def grads_of(gs):
    torch.cuda.synchronize()
    out = {k: v.clone() for (k, _), v in zip([(n, p) for n, p in named], gs.grads.views)} if False else {}
    names = {id(p): n for n, p in named}
    out = {names[id(p)]: v.clone() for p, v in zip(gs.params, gs.grads.views)}
    return out
TR.set_two_streams(False)
one = GraphedStep(model, fwd, cot, params, inputs=gi + gp + [gq])
one._clear_grads(); ref_out = one._fwd_bwd().detach().clone(); ref = grads_of(one)
TR.set_two_streams(os.environ.get('UBV_TWO_STREAMS', '1') != '0')
two = GraphedStep(model, fwd, cot, params, inputs=gi + gp + [gq])
two.capture()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    model.forced_flags = (1, 1)
    two.step() if hasattr(two, 'step') else None
    got = grads_of(two)
    fo = float((two.out.detach() - ref_out).norm() / ref_out.norm())
    d = {k: float((got[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-30)) for k in ref if k in got}
    top = sorted(d.items(), key=lambda kv: -kv[1])[:4]
    print(f'replay {i} vs one-stream eager: fused {fo:.1e};', ', '.join(f'{k.replace("_bev_encoder.layers.", ".L").replace("attentions.", "a")}: {v:.1e}' for k, v in top))
