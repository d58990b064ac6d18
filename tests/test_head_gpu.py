"""Row f1: the consumer of the hot path — ``DetectionTransformerDecoder`` with
``CustomMSDeformableAttention`` and ``UniBEV_Head.forward`` — against vectors recorded from the
reference's own decoder and head (tests/golden/make_golden.py::gen_head; reference:
models/modules/decoder.py:51-338, models/dense_heads/unibev_head.py:145-242)."""
import json

import numpy as np
import pytest
import torch

from _util import checksum, golden, metas_from, t
from unibev_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build_case(name):
    import make_golden as mg          # case table only; does not touch /root/reference
    from unibev_amd.registry import HEADS
    g = golden('head_' + name)
    cfg = json.loads(str(g['cfg_json']))
    case = mg.HEAD_CASES[name]
    img, pts, metas = mg.head_inputs(name, *case)
    np.testing.assert_array_equal(checksum(img[0]), g['img_ck'])
    np.testing.assert_array_equal(checksum(pts[0]), g['pts_ck'])
    head = HEADS.build(json.loads(json.dumps(cfg))).to(DEV).eval()
    named = [(n, tuple(json.loads(s))) for n, s in zip(g['param_names'], g['param_shapes'])]
    sd = syn.seeded_state_dict(named, case[-1])
    missing, unexpected = head.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)      # same state-dict names as the reference
    return head, g, img, pts, metas


@pytest.mark.parametrize('name', ['cnw', 'cat', 'dual'])
def test_head_and_decoder_vs_reference_vectors(name):
    head, g, img, pts, metas = build_case(name)
    got = {}
    head.transformer.register_forward_hook(lambda m, i, o: got.__setitem__('t', o))
    with torch.no_grad():
        outs = head([t(x, device=DEV) for x in img], [t(x, device=DEV) for x in pts], metas)
    _, hs, init_ref, inter_ref = got['t']

    def close(a, ref, tol=1e-4):
        a = a.float().cpu().numpy()
        assert a.shape == ref.shape, (a.shape, ref.shape)
        np.testing.assert_allclose(a, ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max())))

    close(outs['bev_embed'], g['bev_embed'])
    close(init_ref, g['init_reference'])
    close(hs, g['hs'], 2e-4)
    close(inter_ref, g['inter_references'])
    close(outs['all_cls_scores'], g['all_cls_scores'], 2e-4)
    close(outs['all_bbox_preds'], g['all_bbox_preds'], 2e-4)
    assert outs['enc_cls_scores'] is None and outs['enc_bbox_preds'] is None


def test_head_backward_reaches_decoder_and_encoder():
    """Training-style pass through head + decoder: gradients reach the object queries, the decoder's
    sampling layers and the BEV encoder; the decoder's cross-attention (900-query style, arbitrary
    reference points on the fused BEV map) runs its backward on the bins plan, not on atomics, and is
    repeatable bit for bit."""
    head, g, img, pts, metas = build_case('cnw')
    grads = []
    for _ in range(2):
        head.zero_grad(set_to_none=True)
        outs = head([t(x, device=DEV) for x in img], [t(x, device=DEV) for x in pts], metas)
        (outs['all_cls_scores'].square().mean() + outs['all_bbox_preds'].square().mean()).backward()
        need = ['query_embedding.weight', 'bev_embedding.weight',
                'transformer.decoder.layers.0.attentions.1.sampling_offsets.weight',
                'transformer.decoder.layers.1.attentions.0.attn.in_proj_weight',
                'transformer.img_bev_encoder.layers.0.attentions.1.deformable_attention.value_proj.weight',
                'reg_branches.0.4.weight', 'cls_branches.1.6.bias']
        sd = dict(head.named_parameters())
        for n in need:
            assert sd[n].grad is not None and torch.isfinite(sd[n].grad).all() and sd[n].grad.abs().sum() > 0, n
        grads.append(sd['transformer.decoder.layers.0.attentions.1.value_proj.weight'].grad.clone())
    assert torch.equal(grads[0], grads[1])


def test_positional_term_folded_into_offset_gemm_matches_unfolded():
    """The BEV self-attentions add ``bev_pos`` as a per-query bias of their offset / logit GEMM (one table GEMM per
    encoder, encoders._EncoderBase._fold_pos_terms) instead of forming ``query + query_pos``: same outputs and the
    same gradients — queries, positional embeddings, layer weights — as the unfolded path (train mode off: no
    dropout, so the two passes are comparable)."""
    from unibev_amd.modules import encoders as E
    head, g, img, pts, metas = build_case('cnw')
    assert E._FOLD_POS
    names = ['bev_embedding.weight', 'positional_encoding.row_embed.weight', 'positional_encoding.col_embed.weight',
             'transformer.img_bev_encoder.layers.0.attentions.0.sampling_offsets.weight',
             'transformer.img_bev_encoder.layers.0.attentions.0.attention_weights.bias',
             'transformer.pts_bev_encoder.layers.0.attentions.0.attention_weights.weight',
             'transformer.pts_bev_encoder.layers.0.attentions.0.value_proj.weight',
             'transformer.img_bev_encoder.layers.0.attentions.0.output_proj.weight']
    names = [n for n in names if n in dict(head.named_parameters())] + \
        [n for n, _ in head.named_parameters() if n.endswith('layers.1.attentions.0.sampling_offsets.weight')]
    params = dict(head.named_parameters())
    # count which way the self-attentions went: the folded path hands a row_bias to offsets_and_logits
    from unibev_amd.modules.deform_attn import MultiScaleDeformableAttention as MSDA
    calls = {'fold': 0, 'plain': 0, 'layers': 0}
    calls['layers'] = sum(len(enc.layers) for enc in (head.transformer.img_bev_encoder, head.transformer.pts_bev_encoder))
    orig = MSDA.offsets_and_logits

    def counted(self, query, passthru=False, row_bias=None):
        if query.shape[1] == head.bev_h * head.bev_w:           # the encoders' self-attentions, not the decoder's
            calls['fold' if row_bias is not None else 'plain'] += 1
        return orig(self, query, passthru=passthru, row_bias=row_bias)
    MSDA.offsets_and_logits = counted
    # ... or goes straight to the fused value | offsets | logits GEMM
    from unibev_amd.modules import deform_attn as DA
    fused = DA.self_attn_in

    def counted_fused(x, row_bias, *a, **k):
        calls['fold'] += 1
        return fused(x, row_bias, *a, **k)
    DA.self_attn_in = counted_fused
    res = []
    for fold in (True, False):
        E._FOLD_POS = fold
        try:
            head.zero_grad(set_to_none=True)
            outs = head([t(x, device=DEV) for x in img], [t(x, device=DEV) for x in pts], metas)
            loss = outs['bev_embed'].square().mean() + outs['all_bbox_preds'].square().mean()
            loss.backward()
            res.append((outs['bev_embed'].detach().clone(), {n: params[n].grad.clone() for n in names}))
        finally:
            E._FOLD_POS = True
    MSDA.offsets_and_logits = orig
    DA.self_attn_in = fused
    assert calls['fold'] == calls['layers'] > 0 and calls['plain'] == calls['layers'], calls    # one pass each way
    (a, ga), (b, gb) = res
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    for n in names:
        scale = float(gb[n].abs().max())
        assert scale > 0, n
        torch.testing.assert_close(ga[n], gb[n], rtol=2e-3, atol=2e-3 * scale, msg=lambda m, n=n: f'{n}: {m}')
