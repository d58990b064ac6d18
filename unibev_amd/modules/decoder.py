"""``DetectionTransformerDecoder`` — the object-query decoder that consumes ``fused_bev_embed``
(reference: models/modules/decoder.py:51-128).  Every layer is a
``DetrTransformerDecoderLayer``: nn.MultiheadAttention self-attention over the object queries, then
``CustomMSDeformableAttention`` cross-attention in which the num_query (900) queries sample the
bev_h x bev_w fused BEV map around their 2-D reference points — the same fused lifting kernels as
the encoder, with an arbitrary (non-grid) query set.

Parity with the reference is pinned by ``tests/golden/head_*.npz`` (recorded from the reference's
own decoder + head) in ``tests/test_head_gpu.py``.
"""
import torch

from ..registry import TRANSFORMER_LAYER_SEQUENCE
from .bricks import TransformerLayerSequence


def inverse_sigmoid(x, eps=1e-5):
    """[ext] mmdet ``inverse_sigmoid``: logit of x clamped to [eps, 1 - eps]."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def refine_reference(reference_points, box_delta):
    """Iterative box refinement step (decoder.py:103-117): the regression branch's centre outputs
    (x, y in channels 0:2, z in channel 4) move the (bs, nq, 3) reference points in logit space;
    the result is detached, so later layers do not back-propagate into earlier boxes."""
    logit = inverse_sigmoid(reference_points)
    moved = torch.cat((box_delta[..., 0:2] + logit[..., 0:2], box_delta[..., 4:5] + logit[..., 2:3]), -1)
    return moved.sigmoid().detach()


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetectionTransformerDecoder(TransformerLayerSequence):
    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.fp16_enabled = False

    def forward(self, query, *args, reference_points=None, reg_branches=None,
                key_padding_mask=None, **kwargs):
        """query (num_query, bs, C), reference_points (bs, num_query, 3) in [0, 1] ->
        (states (L, num_query, bs, C), references (L, bs, num_query, 3)) when
        ``return_intermediate`` else the last state and reference."""
        states, references = [], []
        x = query
        for depth, layer in enumerate(self.layers):
            # one feature level: (bs, nq, 1, 2) sampling centres in BEV-normalised (x, y)
            x = layer(x, *args, reference_points=reference_points[..., None, :2],
                      key_padding_mask=key_padding_mask, **kwargs)
            if reg_branches is not None:
                assert reference_points.shape[-1] == 3
                reference_points = refine_reference(reference_points,
                                                    reg_branches[depth](x.transpose(0, 1)))
            if self.return_intermediate:
                states.append(x)
                references.append(reference_points)
        if self.return_intermediate:
            return torch.stack(states), torch.stack(references)
        return x, reference_points
