#!/usr/bin/env python3
"""Full-size (cfg4 shapes, bs=1) distance of each precision mode to the REFERENCE-recorded output
subsample (tests/golden/encoder_fullsize.npz).  Prints one JSON line per mode."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
from _util import encoder_case, t                      # noqa: E402
from unibev_amd import build_transformer                # noqa: E402

DEV = 'cuda:0'


def main():
    for fx in (sys.argv[1:] or ['fullsize', 'fullsize_init', 'fullsize_cat128']):
        study(fx)


def study(fixture):
    cfg, sd, inp, g = encoder_case(fixture)
    model = build_transformer(json.loads(json.dumps(cfg))).to(DEV).eval()
    model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    idx, sub = g['fused_idx'], g['fused_sub']
    scale = np.abs(sub).max()
    modes = [('fp32', torch.float32, False, {})]
    modes.append(('fp32/value-fp16', torch.float32, False, {'_value': torch.float16}))
    modes.append(('fp32/value-bf16', torch.float32, False, {'_value': torch.bfloat16}))
    for dt, name in ((torch.float16, 'fp16'), (torch.bfloat16, 'bf16')):
        modes.append((name + '/stream16', dt, True, {}))
        modes.append((name + '/stream32', dt, False, {}))
        modes.append((name + '/stream32/offlog32', dt, False, {'UBV_OFFLOG': 'fp32'}))
        modes.append((name + '/stream16/offlog32', dt, True, {'UBV_OFFLOG': 'fp32'}))
    for name, dt, lowp, env in modes:
        for k in ('UBV_OFFLOG', 'UBV_GEMM_EMU'):
            os.environ.pop(k, None)
        env = dict(env)
        from unibev_amd.modules.deform_attn import set_value_storage
        set_value_storage(env.pop('_value', None))
        os.environ.update(env)
        model.lowp_stream = lowp
        with torch.no_grad(), torch.autocast('cuda', dtype=dt, enabled=dt != torch.float32):
            fused = model.encode([t(x, device=DEV) for x in inp['img']],
                                 [t(x, device=DEV) for x in inp['pts']], t(inp['bev_q'], device=DEV),
                                 inp['bev_h'], inp['bev_w'], bev_pos=t(inp['bev_pos'], device=DEV),
                                 img_metas=inp['metas'])
        f = fused.float().cpu().numpy().reshape(-1)[idx]
        print(json.dumps({'fixture': fixture, 'mode': name, 'normwise': float(np.linalg.norm(f - sub) / np.linalg.norm(sub)),
                          'max_abs_over_max': float(np.abs(f - sub).max() / scale),
                          'p99_abs_over_max': float(np.quantile(np.abs(f - sub), 0.99) / scale)}), flush=True)


if __name__ == '__main__':
    main()
