"""``DetectionTransformerDecoder`` — the object-query decoder that consumes ``fused_bev_embed``
(reference: models/modules/decoder.py:51-128).  It is a consumer of the hot path (SURVEY.md
section 8(f) row f1): each of its layers runs nn.MultiheadAttention self-attention and
``CustomMSDeformableAttention`` cross-attention, i.e. the k1 / bev_lift kernels with Nq = 900
queries sampling the 200x200 fused BEV map.
"""
import torch

from ..registry import TRANSFORMER_LAYER_SEQUENCE
from .bricks import TransformerLayerSequence


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class DetectionTransformerDecoder(TransformerLayerSequence):
    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.fp16_enabled = False

    def forward(self, query, *args, reference_points=None, reg_branches=None,
                key_padding_mask=None, **kwargs):
        """query (num_query, bs, C); reference_points (bs, num_query, 3) in [0, 1].  With
        ``reg_branches`` the reference points are refined after every layer (box refinement) and
        detached, as in the reference."""
        output = query
        intermediate, intermediate_reference_points = [], []
        for lid, layer in enumerate(self.layers):
            reference_points_input = reference_points[..., :2].unsqueeze(2)
            output = layer(output, *args, reference_points=reference_points_input,
                           key_padding_mask=key_padding_mask, **kwargs)
            output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = reg_branches[lid](output)
                assert reference_points.shape[-1] == 3
                new_reference_points = torch.zeros_like(reference_points)
                new_reference_points[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points[..., :2])
                new_reference_points[..., 2:3] = tmp[..., 4:5] + inverse_sigmoid(reference_points[..., 2:3])
                reference_points = new_reference_points.sigmoid().detach()
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points
