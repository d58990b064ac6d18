"""ORACLE — test infrastructure only.  NOT part of the product path.

A CPU (torch fp32 / fp64) restatement of the reference's BEV-encoder hot path, written
functionally over a flat state dict that uses the reference's parameter names
(SURVEY.md Appendix B).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package ``unibev_amd`` never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
golden vectors recorded in this container by importing the reference's own modules from
``/root/reference`` (``tests/golden/make_golden.py``); the bilinear core is additionally pinned
to ``torch.nn.functional.grid_sample`` (the published definition of mmcv-full 1.3.17
``multi_scale_deformable_attn_pytorch``) and to the plain-C loop restatement in
``oracle/msda_ref.c``.

Reference paths are relative to ``/root/reference/projects/UniBEV/unibev_plugin/models/modules/``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- k1: the sampling op
def msda(value, spatial_shapes, sampling_locations, attention_weights):
    """[ext] mmcv ``multi_scale_deformable_attn_pytorch`` (call sites:
    spatial_cross_attention_img.py:437-438, spatial_cross_attention_pts.py:444-445,
    decoder.py:329-330).

    value (B,S,H,Dh); spatial_shapes (L,2) (h,w); sampling_locations (B,Nq,H,L,P,2) as (x,y) in
    [0,1]; attention_weights (B,Nq,H,L,P)  ->  (B,Nq,H*Dh).
    """
    B, _, H, Dh = value.shape
    _, Nq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in np.asarray(spatial_shapes).reshape(-1, 2)]
    vals = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = vals[lvl].flatten(2).transpose(1, 2).reshape(B * H, Dh, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(B * H, 1, Nq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1)
    return out.view(B, H * Dh, Nq).transpose(1, 2).contiguous()


# --------------------------------------------------------------------------- reference points
def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim='3d', bs=1,
                         dtype=torch.float32):
    """encoder_unibev_detr_img.py:45-109 (identical in encoder_unibev_detr_pts.py:45-102)."""
    if dim == '3d':
        zs = torch.linspace(0.5, Z - 0.5, num_points_in_pillar, dtype=dtype
                            ).view(-1, 1, 1).expand(num_points_in_pillar, H, W) / Z
        xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype
                            ).view(1, 1, W).expand(num_points_in_pillar, H, W) / W
        ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype
                            ).view(1, H, 1).expand(num_points_in_pillar, H, W) / H
        ref_3d = torch.stack((xs, ys, zs), -1)
        ref_3d = ref_3d.permute(0, 3, 1, 2).flatten(2).permute(0, 2, 1)
        return ref_3d[None].repeat(bs, 1, 1, 1)                       # (bs, D, Nq, 3)
    ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype),
                                  torch.linspace(0.5, W - 0.5, W, dtype=dtype), indexing='ij')
    ref_y = ref_y.reshape(-1)[None] / H
    ref_x = ref_x.reshape(-1)[None] / W
    ref_2d = torch.stack((ref_x, ref_y), -1)
    return ref_2d.repeat(bs, 1, 1).unsqueeze(2)                       # (bs, Nq, 1, 2)


def point_sampling_img(reference_points, pc_range, img_metas):
    """encoder_unibev_detr_img.py:112-187.  Returns reference_points_cam (Nc,B,Nq,D,2) and
    bev_mask (Nc,B,Nq,D) bool.  Quirk q5: img_shape of sample 0 normalises the whole batch."""
    lidar2img = np.asarray([m['lidar2img'] for m in img_metas])
    lidar2img = reference_points.new_tensor(lidar2img)                # (B, Nc, 4, 4)
    rp = reference_points.clone()
    rp[..., 0:1] = rp[..., 0:1] * (pc_range[3] - pc_range[0]) + pc_range[0]
    rp[..., 1:2] = rp[..., 1:2] * (pc_range[4] - pc_range[1]) + pc_range[1]
    rp[..., 2:3] = rp[..., 2:3] * (pc_range[5] - pc_range[2]) + pc_range[2]
    rp = torch.cat((rp, torch.ones_like(rp[..., :1])), -1)
    rp = rp.permute(1, 0, 2, 3)                                       # (D, B, Nq, 4)
    D, B, Nq = rp.shape[:3]
    Nc = lidar2img.size(1)
    rp = rp.view(D, B, 1, Nq, 4).repeat(1, 1, Nc, 1, 1).unsqueeze(-1)
    l2i = lidar2img.view(1, B, Nc, 1, 4, 4).repeat(D, 1, 1, Nq, 1, 1)
    cam = torch.matmul(l2i.to(torch.float32), rp.to(torch.float32)).squeeze(-1)
    eps = 1e-5
    mask = cam[..., 2:3] > eps
    cam = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
    cam[..., 0] /= img_metas[0]['img_shape'][0][1]
    cam[..., 1] /= img_metas[0]['img_shape'][0][0]
    mask = (mask & (cam[..., 1:2] > 0.0) & (cam[..., 1:2] < 1.0)
            & (cam[..., 0:1] < 1.0) & (cam[..., 0:1] > 0.0))
    mask = torch.nan_to_num(mask)
    cam = cam.permute(2, 1, 3, 0, 4)
    mask = mask.permute(2, 1, 3, 0, 4).squeeze(-1)
    return cam, mask


def point_sampling_pts(reference_points):
    """encoder_unibev_detr_pts.py:105-127: (bs,D,Nq,3) -> (D,bs,Nq,2); the mask is discarded
    by the caller (:169), quirk q6."""
    return reference_points.clone().permute(1, 0, 2, 3)[..., :2]


# --------------------------------------------------------------------------- attention modules
def _lin(P, name, x):
    return F.linear(x, P[name + '.weight'], P[name + '.bias'])


def msda3d(P, pre, query, value, reference_points, spatial_shapes, num_heads=8, num_levels=1,
           num_points=8):
    """MSDeformableAttention3DImg.forward / MSDeformableAttention3DPts.forward
    (spatial_cross_attention_img.py:313-442, spatial_cross_attention_pts.py:306-449);
    batch_first, no output_proj, no residual.  Quirk q3: flat point p uses Z-anchor p % Z."""
    bs, nq, _ = query.shape
    _, nv, _ = value.shape
    ss = torch.as_tensor(np.asarray(spatial_shapes).reshape(-1, 2), dtype=torch.long)
    assert int((ss[:, 0] * ss[:, 1]).sum()) == nv
    v = _lin(P, pre + 'value_proj', value).view(bs, nv, num_heads, -1)
    off = _lin(P, pre + 'sampling_offsets', query).view(bs, nq, num_heads, num_levels,
                                                        num_points, 2)
    aw = _lin(P, pre + 'attention_weights', query).view(bs, nq, num_heads,
                                                        num_levels * num_points)
    aw = aw.softmax(-1).view(bs, nq, num_heads, num_levels, num_points)
    normalizer = torch.stack([ss[..., 1], ss[..., 0]], -1).to(query.dtype)
    Z = reference_points.shape[2]
    rp = reference_points[:, :, None, None, None, :, :]
    off = off / normalizer[None, None, None, :, None, :]
    off = off.view(bs, nq, num_heads, num_levels, num_points // Z, Z, 2)
    loc = (rp + off).view(bs, nq, num_heads, num_levels, num_points, 2)
    return msda(v, ss, loc, aw)


def self_msda(P, pre, query, query_pos, reference_points, spatial_shapes, num_heads=8,
              num_levels=1, num_points=4):
    """[ext] mmcv MultiScaleDeformableAttention in the self-attn slot == the vendored copy
    decoder.py:278-338 (batch_first, eval): value = identity = query (without pos), quirk q4."""
    bs, nq, _ = query.shape
    identity = query
    value = query
    q = query + query_pos if query_pos is not None else query
    ss = torch.as_tensor(np.asarray(spatial_shapes).reshape(-1, 2), dtype=torch.long)
    v = _lin(P, pre + 'value_proj', value).view(bs, nq, num_heads, -1)
    off = _lin(P, pre + 'sampling_offsets', q).view(bs, nq, num_heads, num_levels, num_points, 2)
    aw = _lin(P, pre + 'attention_weights', q).view(bs, nq, num_heads, num_levels * num_points)
    aw = aw.softmax(-1).view(bs, nq, num_heads, num_levels, num_points)
    normalizer = torch.stack([ss[..., 1], ss[..., 0]], -1).to(query.dtype)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = _lin(P, pre + 'output_proj', msda(v, ss, loc, aw))
    return out + identity


def sca_img(P, pre, query, value, reference_points_cam, bev_mask, spatial_shapes,
            num_cams, num_points=8, num_levels=1):
    """SpatialCrossAttentionImg.forward (spatial_cross_attention_img.py:67-215), eval mode.
    Quirk q1: re-batch indices come from batch element 0; q2: count is per batch element."""
    inp_residual = query
    slots = torch.zeros_like(query)
    bs, nq, C = query.shape
    D = reference_points_cam.size(3)
    indexes = [m[0].sum(-1).nonzero().squeeze(-1) for m in bev_mask]
    max_len = max(len(i) for i in indexes)
    q_rb = query.new_zeros(bs, num_cams, max_len, C)
    r_rb = reference_points_cam.new_zeros(bs, num_cams, max_len, D, 2)
    for j in range(bs):
        for i in range(num_cams):
            idx = indexes[i]
            q_rb[j, i, :len(idx)] = query[j, idx]
            r_rb[j, i, :len(idx)] = reference_points_cam[i][j, idx]
    _, l, _, _ = value.shape
    val = value.permute(2, 0, 1, 3).reshape(bs * num_cams, l, C)
    out = msda3d(P, pre + 'deformable_attention.', q_rb.view(bs * num_cams, max_len, C), val,
                 r_rb.view(bs * num_cams, max_len, D, 2), spatial_shapes,
                 num_levels=num_levels, num_points=num_points)
    out = out.view(bs, num_cams, max_len, C)
    for j in range(bs):
        for i, idx in enumerate(indexes):
            slots[j, idx] += out[j, i, :len(idx)]
    count = (bev_mask.sum(-1) > 0).permute(1, 2, 0).sum(-1)
    count = torch.clamp(count, min=1.0)
    slots = slots / count[..., None]
    return _lin(P, pre + 'output_proj', slots) + inp_residual


def sca_pts(P, pre, query, value, reference_points_lidar, spatial_shapes, num_points=8,
            num_levels=1):
    """SpatialCrossAttentionPts.forward (spatial_cross_attention_pts.py:64-206), eval mode."""
    val = value.permute(1, 0, 2)
    rp = reference_points_lidar.permute(1, 2, 0, 3)
    out = msda3d(P, pre + 'deformable_attention.', query, val, rp, spatial_shapes,
                 num_levels=num_levels, num_points=num_points)
    return _lin(P, pre + 'output_proj', out) + query


def ffn(P, pre, x):
    """[ext] mmcv FFN (num_fcs=2, ReLU, add_identity), eval mode."""
    h = F.relu(_lin(P, pre + 'layers.0.0', x))
    return x + _lin(P, pre + 'layers.1', h)


def layer_norm(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + '.weight'], P[pre + '.bias'], 1e-5)


def encoder_layer(P, pre, kind, query, value, bev_pos, ref_2d, bev_h, bev_w, cross_kwargs):
    """ImgLayer.forward / PtsLayer.forward with operation_order
    ('self_attn','norm','cross_attn','norm','ffn','norm')
    (encoder_unibev_detr_img.py:339-481, encoder_unibev_detr_pts.py:256-355)."""
    q = self_msda(P, pre + 'attentions.0.', query, bev_pos, ref_2d, [[bev_h, bev_w]])
    q = layer_norm(P, pre + 'norms.0', q)
    if kind == 'img':
        q = sca_img(P, pre + 'attentions.1.', q, value, **cross_kwargs)
    else:
        q = sca_pts(P, pre + 'attentions.1.', q, value, **cross_kwargs)
    q = layer_norm(P, pre + 'norms.1', q)
    q = ffn(P, pre + 'ffns.0.', q)
    return layer_norm(P, pre + 'norms.2', q)


def img_encoder(P, pre, bev_query, value, bev_h, bev_w, bev_pos, spatial_shapes, img_metas,
                pc_range, num_layers, num_cams, num_points_in_pillar=4, num_points=8,
                return_aux=False):
    """ImgEncoder.forward (encoder_unibev_detr_img.py:189-289)."""
    bs = bev_query.size(1)
    ref_3d = get_reference_points(bev_h, bev_w, pc_range[5] - pc_range[2], num_points_in_pillar,
                                  '3d', bs, bev_query.dtype)
    ref_2d = get_reference_points(bev_h, bev_w, dim='2d', bs=bs, dtype=bev_query.dtype)
    cam, mask = point_sampling_img(ref_3d, pc_range, img_metas)
    q = bev_query.permute(1, 0, 2)
    pos = bev_pos.permute(1, 0, 2) if bev_pos is not None else None
    ck = dict(reference_points_cam=cam, bev_mask=mask, spatial_shapes=spatial_shapes,
              num_cams=num_cams, num_points=num_points)
    for i in range(num_layers):
        q = encoder_layer(P, f'{pre}layers.{i}.', 'img', q, value, pos, ref_2d, bev_h, bev_w, ck)
    return (q, cam, mask) if return_aux else q


def pts_encoder(P, pre, bev_query, value, bev_h, bev_w, bev_pos, spatial_shapes, pc_range,
                num_layers, num_points_in_pillar_lidar=4, num_points=8):
    """PtsEncoder.forward (encoder_unibev_detr_pts.py:129-209)."""
    bs = bev_query.size(1)
    ref_3d = get_reference_points(bev_h, bev_w, pc_range[5] - pc_range[2],
                                  num_points_in_pillar_lidar, '3d', bs, bev_query.dtype)
    ref_2d = get_reference_points(bev_h, bev_w, dim='2d', bs=bs, dtype=bev_query.dtype)
    rpl = point_sampling_pts(ref_3d)
    q = bev_query.permute(1, 0, 2)
    pos = bev_pos.permute(1, 0, 2) if bev_pos is not None else None
    ck = dict(reference_points_lidar=rpl, spatial_shapes=spatial_shapes, num_points=num_points)
    for i in range(num_layers):
        q = encoder_layer(P, f'{pre}layers.{i}.', 'pts', q, value, pos, ref_2d, bev_h, bev_w, ck)
    return q


# --------------------------------------------------------------------------- transformer glue
def pre_process_img_feats(P, mlvl_img_feats, use_cams_embeds=True):
    """UniBEVTransformer._pre_process_img_feats (transformer_fusion.py:230-255)."""
    flat, shapes = [], []
    for lvl, feat in enumerate(mlvl_img_feats):
        bs, nc, c, h, w = feat.shape
        feat = feat.flatten(3).permute(1, 0, 3, 2)
        if use_cams_embeds:
            feat = feat + P['cams_embeds'][:, None, None, :].to(feat.dtype)
        feat = feat + P['img_level_embeds'][None, None, lvl:lvl + 1, :].to(feat.dtype)
        shapes.append((h, w))
        flat.append(feat)
    flat = torch.cat(flat, 2).permute(0, 2, 1, 3)                     # (Nc, sum hw, bs, C)
    return flat, shapes


def pre_process_pts_feats(P, mlvl_pts_feats):
    """UniBEVTransformer._pre_process_pts_feats (transformer_fusion.py:257-278); the level
    concat on the channel dim (:272) is kept — harmless at one level."""
    flat, shapes = [], []
    for lvl, feat in enumerate(mlvl_pts_feats):
        bs, c, h, w = feat.shape
        feat = feat.flatten(2).permute(0, 2, 1)
        feat = feat + P['pts_level_embeds'][None, lvl:lvl + 1, :].to(feat.dtype)
        shapes.append((h, w))
        flat.append(feat)
    return torch.cat(flat, 2).permute(1, 0, 2), shapes                # (hw, bs, C)


_MLP_NORM_ACT = {
    'MLP_ChannelNormWeights': torch.relu,
    'Leaky_ReLU_MLP_ChannelNormWeights': lambda v: F.leaky_relu(v, 0.01),
    'ELU_MLP_ChannelNormWeights': F.elu,
    'Sigmoid_MLP_ChannelNormWeights': torch.sigmoid,
}


def modality_projection(P, prefix, x):
    """ModalityProjectionModule.forward (transformer_fusion.py:26-47): x + LayerNorm(relu(Linear(x)))."""
    h = torch.relu(F.linear(x, P[prefix + 'net.0.weight'], P[prefix + 'net.0.bias']))
    h = F.layer_norm(h, (h.shape[-1],), P[prefix + 'net.2.weight'], P[prefix + 'net.2.bias'])
    return x + h


def channel_feature_norm(P, img, pts, feature_norm, c_flag, l_flag):
    """transformer_fusion.py:316-384: ChannelNormWeights (:323-337), the learned per-sample variants
    (:345-361) and the modality projection (:369-374)."""
    if img is None:
        img = torch.zeros_like(pts)
    elif pts is None:
        pts = torch.zeros_like(img)
    if feature_norm == 'ChannelNormWeights':
        fw = torch.cat([P['img_channel_weights'][None], P['pts_channel_weights'][None]], 0)
        if c_flag == 1 and l_flag == 1:
            n = fw.softmax(0)
            iw, pw = n[0], n[1]
        else:
            iw = fw[0:1].softmax(0)[0]
            pw = fw[1:2].softmax(0)[0]
        img = img * iw
        pts = pts * pw
    elif feature_norm in _MLP_NORM_ACT:
        # one 2-way score per (sample, channel) from ALL tokens of both modalities: Linear over the token axis
        tokens = torch.cat([img, pts], 1).transpose(1, 2)                      # (bs, C, 2 Nq)
        score = _MLP_NORM_ACT[feature_norm](F.linear(tokens, P['channel_weights_proj.0.weight'],
                                                     P['channel_weights_proj.0.bias']))   # (bs, C, 2)
        if c_flag == 1 and l_flag == 1:
            n = score.softmax(-1)
            iw, pw = n[..., 0], n[..., 1]
        else:
            iw = score[..., :1].softmax(-1)[..., 0]                            # softmax of one entry: ones
            pw = score[..., 1:].softmax(-1)[..., 0]
        img = img * iw[:, None, :]
        pts = pts * pw[:, None, :]
    elif feature_norm == 'ModalityProjection':
        pseudo_pts = modality_projection(P, 'l_modal_proj.', img)
        pseudo_img = modality_projection(P, 'c_modal_proj.', pts)
        img = torch.cat([img, pseudo_pts], -1)
        pts = torch.cat([pseudo_img, pts], -1)
    elif feature_norm is not None:
        raise NotImplementedError(feature_norm)
    return img, pts


def spatial_feature_norm(P, img, pts, spatial_norm, c_flag, l_flag):
    """transformer_fusion.py:386-413."""
    if spatial_norm == 'SpatialNormWeights':
        sw = torch.cat([P['img_spatial_weights'][None], P['pts_spatial_weights'][None]], 0)
        if c_flag == 1 and l_flag == 1:
            n = sw.softmax(0)
            iw, pw = n[0], n[1]
        else:
            iw = sw[:1].softmax(0)[0]
            pw = sw[1:].softmax(0)[0]
        img = img * iw[None, :, None]
        pts = pts * pw[None, :, None]
    return img, pts


def multi_modal_fusion(img, pts, fusion_method, c_flag, l_flag, feature_norm=None, P=None,
                       use_modal_embeds=None):
    """transformer_fusion.py:280-314: linear / avg / cat, the flag vectors of the modality projection
    (:287-300) and the modal embeddings (:304-310)."""
    if fusion_method == 'linear':
        fused = c_flag * img + l_flag * pts
    elif fusion_method == 'avg':
        fused = img * c_flag / (c_flag + l_flag) + pts * l_flag / (c_flag + l_flag)
    elif fusion_method == 'cat':
        if feature_norm == 'ModalityProjection':
            # img = [img | pseudo pts], pts = [pseudo img | pts]: a half is the real feature when its
            # modality is present and the projection of the other modality when it is not
            C = img.shape[-1] // 2
            img_flags = torch.cat([torch.full((C,), float(c_flag)), torch.full((C,), float(1 - l_flag))]).to(img)
            pts_flags = torch.cat([torch.full((C,), float(1 - c_flag)), torch.full((C,), float(l_flag))]).to(img)
            fused = img * img_flags + pts * pts_flags
        else:
            fused = torch.cat((img * c_flag, pts * l_flag), -1)
    else:
        raise ValueError('Unrecognizable fusion method:{}'.format(fusion_method))
    if use_modal_embeds == 'MLP':
        status = torch.tensor([float(c_flag), float(l_flag)]).to(fused)
        h = torch.relu(F.linear(status, P['modal_embbeding_mlp.0.weight'], P['modal_embbeding_mlp.0.bias']))
        fused = fused + torch.relu(F.linear(h, P['modal_embbeding_mlp.2.weight'], P['modal_embbeding_mlp.2.bias']))
    elif use_modal_embeds == 'Fixed':
        fused = fused + (c_flag * P['modal_embbeding_C'] + l_flag * P['modal_embbeding_L'])
    return fused


def transformer_encode_fuse(P, cfg, img_mlvl_feats, pts_mlvl_feats, bev_queries, bev_h, bev_w,
                            bev_pos, img_metas, c_flag=1, l_flag=1, return_parts=False):
    """UniBEVTransformer.forward up to ``fused_bev_embed`` (transformer_fusion.py:463-549),
    eval mode (modality-dropout flags are inputs here).  ``P`` holds the transformer's state
    dict (names as in SURVEY.md Appendix B, without the ``pts_bbox_head.transformer.`` prefix).
    ``cfg`` is the UniBEVTransformer config dict.  Returns (Nq, bs, C*s)."""
    if img_mlvl_feats is None:
        c_flag = 0
        bs = pts_mlvl_feats[0].size(0)
    elif pts_mlvl_feats is None:
        l_flag = 0
        bs = img_mlvl_feats[0].size(0)
    else:
        bs = img_mlvl_feats[0].size(0)
    pos = bev_pos.flatten(2).permute(2, 0, 1) if bev_pos is not None else None
    if cfg.get('dual_queries', False):
        q_img = bev_queries[0].unsqueeze(1).repeat(1, bs, 1)
        q_pts = bev_queries[1].unsqueeze(1).repeat(1, bs, 1)
    else:
        q_img = q_pts = bev_queries.unsqueeze(1).repeat(1, bs, 1)
    img_bev = pts_bev = None
    if img_mlvl_feats is not None:
        ec = cfg['img_encoder']
        da = ec['transformerlayers']['attn_cfgs'][1]['deformable_attention']
        flat, shapes = pre_process_img_feats(P, img_mlvl_feats, cfg.get('use_cams_embeds', True))
        img_bev = img_encoder(P, 'img_bev_encoder.', q_img, flat, bev_h, bev_w, pos, shapes,
                              img_metas, ec['pc_range'], ec['num_layers'],
                              cfg.get('num_cams', 6), ec.get('num_points_in_pillar', 4),
                              da.get('num_points', 8))
    if pts_mlvl_feats is not None:
        ec = cfg['pts_encoder']
        da = ec['transformerlayers']['attn_cfgs'][1]['deformable_attention']
        flat, shapes = pre_process_pts_feats(P, pts_mlvl_feats)
        pts_bev = pts_encoder(P, 'pts_bev_encoder.', q_pts, flat, bev_h, bev_w, pos, shapes,
                              ec['pc_range'], ec['num_layers'],
                              ec.get('num_points_in_pillar_lidar', 1), da.get('num_points', 8))
    parts = (img_bev, pts_bev)
    img_n, pts_n = channel_feature_norm(P, img_bev, pts_bev, cfg.get('feature_norm'), c_flag, l_flag)
    img_n, pts_n = spatial_feature_norm(P, img_n, pts_n, cfg.get('spatial_norm'), c_flag, l_flag)
    fused = multi_modal_fusion(img_n, pts_n, cfg.get('fusion_method', 'linear'), c_flag, l_flag,
                               cfg.get('feature_norm'), P, cfg.get('use_modal_embeds'))
    fused = fused.permute(1, 0, 2)
    return (fused, parts) if return_parts else fused


def learned_positional_encoding(row_embed, col_embed, bs, h, w):
    """[ext] mmdet LearnedPositionalEncoding.forward as called at unibev_head.py:179-182:
    cat(col_embed(x), row_embed(y)) -> (bs, 2F, h, w), x half first."""
    x_embed = col_embed[:w]
    y_embed = row_embed[:h]
    pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1),
                     y_embed.unsqueeze(1).repeat(1, w, 1)), dim=-1)
    return pos.permute(2, 0, 1).unsqueeze(0).repeat(bs, 1, 1, 1)


def state_dict_to_torch(sd, dtype=torch.float32):
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in sd.items()}


def msda_init_bias(num_heads, num_levels, num_points):
    """The sampling_offsets bias initialiser shared by all deformable attentions
    (spatial_cross_attention_img.py:293-307, decoder.py:208-222)."""
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    g = torch.stack([thetas.cos(), thetas.sin()], -1)
    g = (g / g.abs().max(-1, keepdim=True)[0]).view(num_heads, 1, 1, 2).repeat(
        1, num_levels, num_points, 1)
    for i in range(num_points):
        g[:, :, i, :] *= i + 1
    return g.view(-1)
