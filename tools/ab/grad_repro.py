"""Run-to-run reproducibility of the product's own gradients (full-size smooth fixture): python tools/ab/grad_repro.py [runs]
UBV_REPRO_DTYPE=bf16|fp16 runs the passes under autocast (the one-stream reference too)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
import test_modules_gpu as T
from _util import encoder_case, t, tq
from unibev_amd import synthetic as syn
DEV = 'cuda'
cfg, sd, inp, g = encoder_case('fullsize_smooth')
nq, bs = inp['bev_h'] * inp['bev_w'], inp['bs']
cot = t(syn.seeded_array('cot:fullsize_smooth', (nq, bs, cfg['embed_dims']), 5) / nq ** 0.5, device=DEV)
model = T._build(cfg).to(DEV).eval()
T._load(model, sd)
gi = [t(x, device=DEV).requires_grad_() for x in inp['img']]
gp = [t(x, device=DEV).requires_grad_() for x in inp['pts']]
gq = tq(inp['bev_q'], device=DEV, grad=True)
bev_pos = t(inp['bev_pos'], device=DEV)
named = [(k, p) for k, p in model.named_parameters() if not k.startswith('decoder') and not k.startswith('reference_points')]
INTER = {}
def _watch(mod, name):
    def fwd_hook(m, args, out):
        o = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(o) and o.requires_grad:
            o.register_hook(lambda g, n=name: INTER.__setitem__('d ' + n, g.detach().clone()))
    mod.register_forward_hook(fwd_hook)
if os.environ.get('UBV_WATCH', '0') == '1':
    for en in ('img_bev_encoder', 'pts_bev_encoder'):
        enc = getattr(model, en)
        for li, layer in enumerate(enc.layers):
            _watch(layer, f'{en[:3]}.L{li}.out')
            for ai, a in enumerate(layer.attentions):
                _watch(a, f'{en[:3]}.L{li}.a{ai}.out')
            for fi, f in enumerate(layer.ffns):
                _watch(f, f'{en[:3]}.L{li}.ffn.out')
ADT = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(os.environ.get('UBV_REPRO_DTYPE', ''))
def run():
    INTER.clear()
    for x in gi + gp + [gq] + [p for _, p in named]:
        x.grad = None
    with torch.autocast('cuda', dtype=ADT or torch.bfloat16, enabled=ADT is not None):
        fused = model.encode(gi, gp, gq, inp['bev_h'], inp['bev_w'], bev_pos=bev_pos, img_metas=inp['metas'])
    (fused.float() * cot).sum().backward()
    torch.cuda.synchronize()
    out = {'fused': fused.detach().clone(), 'img feats': gi[0].grad.clone(), 'pts feats': gp[0].grad.clone(), 'bev queries': gq.grad.clone()}
    out.update({k: p.grad.clone() for k, p in named if p.grad is not None})
    out.update(INTER)
    return out
from unibev_amd.modules import transformer as TR
TR.set_two_streams(False)
ref = run()          # one stream: the reference (reproducible to ~6e-6)
TR.set_two_streams(os.environ.get('UBV_TWO_STREAMS', '1') != '0')
runs = [run() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3)]
for i, r in enumerate(runs):
    d = {k: float((r[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-30)) for k in ref if k in r}
    bad_inter = [f'{k}: {v:.1e}' for k, v in d.items() if k.startswith('d ') and v > 1e-4]
    if bad_inter: print('   corrupted intermediates:', ', '.join(bad_inter))
    top = sorted(d.items(), key=lambda kv: -kv[1])[:6]
    nz = sum(v > 0 for v in d.values())
    print(f'run {i} vs one-stream: {nz}/{len(d)} tensors differ; fused {d["fused"]:.1e};', ', '.join(f'{k.replace("_bev_encoder.layers.", ".L").replace("attentions.", "a")}: {v:.1e}' for k, v in top))
