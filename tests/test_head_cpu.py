"""Head / decoder construction facts that need no GPU: state-dict names and the initialisation rules
of models/dense_heads/unibev_head.py:86-143, against facts recorded from the reference."""
import json

import numpy as np
import torch

from _util import golden


def test_head_state_dict_names_equal_the_reference():
    from unibev_amd.registry import HEADS
    for name in ('cnw', 'cat', 'dual'):
        g = golden('head_' + name)
        head = HEADS.build(json.loads(str(g['cfg_json'])))
        mine = {k: tuple(v.shape) for k, v in head.state_dict().items()}
        ref = {n: tuple(json.loads(s)) for n, s in zip(g['param_names'], g['param_shapes'])}
        assert mine == ref


def test_head_init_weights_follow_the_reference():
    """unibev_head.py:137-143: focal prior on the classification biases, positional encoding left
    at nn.Embedding's initialisation."""
    import make_golden as mg
    from unibev_amd import configs as cfgs
    from unibev_amd.registry import HEADS
    g = golden('head_init')
    head = HEADS.build(json.loads(json.dumps(cfgs.head_cfg(**mg.HEAD_CASES['cnw'][0], **mg.HEAD_CASES['cnw'][1]))))
    before = head.positional_encoding.row_embed.weight.detach().clone()
    head.init_weights()
    for m in head.cls_branches:
        np.testing.assert_allclose(m[-1].bias.detach().numpy(), g['cls_bias'], rtol=1e-6)
    assert bool(g['pos_untouched'][0]) and torch.equal(head.positional_encoding.row_embed.weight, before)
