#!/usr/bin/env python3
"""What the value-only fp16 mode's distance to the reference vectors is made of: per full-size
fixture, normwise distance of ``fused_bev_embed`` for
  fp32            everything f32
  value-fp16      the product mode: value maps AND sampled outputs stored in fp16 (fp16 kernels)
  round(value)    f32 kernels on value maps rounded through fp16 (the stored map's rounding alone)
  round(out)      f32 kernels, sampled outputs rounded through fp16 (the output's rounding alone)
  round(both)     both roundings on f32 kernels (= value-fp16 up to the kernels' own arithmetic)
and the same with bf16.  Inference, eval mode, bs = 1 (tests/_util.encoder_case)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from _util import encoder_case, t                       # noqa: E402
from unibev_amd import build_transformer                # noqa: E402
from unibev_amd import functional as UF                 # noqa: E402
from unibev_amd.modules.deform_attn import set_value_storage   # noqa: E402

dev = 'cuda'
for fx in ('fullsize_init', 'fullsize', 'fullsize_cat128'):
    cfg, sd, inp, g = encoder_case(fx)
    model = build_transformer(json.loads(json.dumps(cfg))).to(dev).eval()
    model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    img = None if inp['img'] is None else [t(x, device=dev) for x in inp['img']]
    pts = None if inp['pts'] is None else [t(x, device=dev) for x in inp['pts']]

    def dist(store=None, study=None, sdt=torch.float16):
        p1, p2 = set_value_storage(store), UF.set_study_rounding(study, sdt)
        try:
            with torch.no_grad():
                fused = model.encode(img, pts, t(inp['bev_q'], device=dev), inp['bev_h'], inp['bev_w'],
                                     bev_pos=t(inp['bev_pos'], device=dev), img_metas=inp['metas'])
        finally:
            set_value_storage(p1)
            UF.set_study_rounding(*(p2 or (None,)))
        f = fused.float().cpu().numpy().reshape(-1)
        ref = g['fused_sub'] if 'fused_sub' in g else g['fused'].reshape(-1)
        f = f[g['fused_idx']] if 'fused_idx' in g else f
        return float(np.linalg.norm(f - ref) / np.linalg.norm(ref))

    row = {'fp32': dist()}
    for name, dt in (('fp16', torch.float16), ('bf16', torch.bfloat16)):
        row[f'value-{name}'] = dist(store=dt)
        row[f'round(value) {name}'] = dist(study='value', sdt=dt)
        row[f'round(out) {name}'] = dist(study='out', sdt=dt)
        row[f'round(both) {name}'] = dist(study='value+out', sdt=dt)
    print(fx, ' '.join(f'{k}={v:.2e}' for k, v in row.items()), flush=True)
    del model
    torch.cuda.empty_cache()
