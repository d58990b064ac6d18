"""CPU oracle of the sparse 3-D convolutions and of SparseEncoder — TEST
INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

PARITY UNPINNED BY THE REFERENCE: spconv / mmdet3d are not vendored in /root/reference and the reference holds
no vectors for this path.  The oracle restates the PUBLISHED semantics of spconv 1.x / mmdet3d 0.18.1 in an
independent form — dense ``torch.nn.functional.conv3d`` on the densified grid plus an explicit activity mask:

* SubMConv3d: outputs exist exactly at the input's active sites; out = conv3d(dense_in, stride 1, padding k//2)
  read at those sites (inactive inputs contribute zeros — they are zeros in the dense grid).
* SparseConv3d: out = conv3d(dense_in, stride, padding); a site is active iff any input in its receptive field
  is active (conv3d of the 0/1 mask > 0).
* weight layout (kz, ky, kx, Cin, Cout), cross-correlation (``out[o] = sum_k W_k in[o * s - p + k]``), i.e.
  conv3d with weight.permute(4, 3, 0, 1, 2).
* BatchNorm1d over the ACTIVE rows only (training-mode batch statistics, biased variance for normalisation),
  ReLU, SparseBasicBlock residual, SparseEncoder layout — mmdet3d/models/middle_encoders/sparse_encoder.py,
  mmdet3d/ops/sparse_block.py.
"""
import torch
import torch.nn.functional as F


def densify(feats, coors, batch_size, shape):
    """(dense [B, C, D, H, W], mask [B, 1, D, H, W]) of features [N, C] at coors [N, 4] (b, z, y, x)."""
    D, H, W = shape
    C = feats.shape[1]
    c = coors.long()
    flat = ((c[:, 0] * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3]
    dense = torch.zeros(batch_size * D * H * W, C, dtype=feats.dtype).index_add(0, flat, feats)
    mask = torch.zeros(batch_size * D * H * W, dtype=feats.dtype).index_fill(0, flat, 1.0)
    return dense.view(batch_size, D, H, W, C).permute(0, 4, 1, 2, 3), mask.view(batch_size, 1, D, H, W)


def _w(weight):
    return weight.permute(4, 3, 0, 1, 2)


def subm_conv(dense, mask, weight):
    k = weight.shape[:3]
    out = F.conv3d(dense, _w(weight), padding=tuple(s // 2 for s in k))
    return out * mask, mask


def sparse_conv(dense, mask, weight, stride, padding):
    k = weight.shape[:3]
    out = F.conv3d(dense, _w(weight), stride=stride, padding=padding)
    m = (F.conv3d(mask, torch.ones(1, 1, *k, dtype=mask.dtype), stride=stride, padding=padding) > 0).to(mask.dtype)
    return out * m, m


def batch_norm(dense, mask, gamma, beta, eps):
    """Training-mode BatchNorm1d over the active sites."""
    n = mask.sum()
    mean = (dense * mask).sum((0, 2, 3, 4), keepdim=True) / n
    var = (((dense - mean) * mask) ** 2).sum((0, 2, 3, 4), keepdim=True) / n
    out = (dense - mean) / torch.sqrt(var + eps) * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)
    return out * mask


def _t3(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


def sparse_encoder(P, cfg, feats, coors, batch_size, eps=1e-3):
    """mmdet3d SparseEncoder.forward with block_type 'basicblock' or 'conv_module', order (conv, norm, act);
    ``P``: state-dict-keyed tensors, ``cfg``: the config kwargs."""
    x, m = densify(feats, coors, batch_size, cfg['sparse_shape'])

    def cna(prefix, x, m, conv, **kw):
        x, m = conv(x, m, P[prefix + '.0.weight'], **kw)
        x = batch_norm(x, m, P[prefix + '.1.weight'], P[prefix + '.1.bias'], eps)
        return torch.relu(x) * m, m

    x, m = cna('conv_input', x, m, subm_conv)
    chans, pads = cfg['encoder_channels'], cfg['encoder_paddings']
    basic = cfg.get('block_type', 'conv_module') == 'basicblock'
    for i, blocks in enumerate(chans):
        for j, _ in enumerate(blocks):
            pre = f'encoder_layers.encoder_layer{i + 1}.{j}'
            pad = _t3(pads[i][j])
            if (not basic) and i != 0 and j == 0:
                x, m = cna(pre, x, m, sparse_conv, stride=(2, 2, 2), padding=pad)
            elif basic and j == len(blocks) - 1 and i != len(chans) - 1:
                x, m = cna(pre, x, m, sparse_conv, stride=(2, 2, 2), padding=pad)
            elif basic:
                idt = x
                y, _ = subm_conv(x, m, P[pre + '.conv1.weight'])
                y = torch.relu(batch_norm(y, m, P[pre + '.bn1.weight'], P[pre + '.bn1.bias'], eps)) * m
                y, _ = subm_conv(y, m, P[pre + '.conv2.weight'])
                y = batch_norm(y, m, P[pre + '.bn2.weight'], P[pre + '.bn2.bias'], eps)
                x = torch.relu(y + idt) * m
            else:
                x, m = cna(pre, x, m, subm_conv)
    x, m = cna('conv_out', x, m, sparse_conv, stride=(2, 1, 1), padding=(0, 0, 0))
    B, C, D, H, W = x.shape
    return x.reshape(B, C * D, H, W)
