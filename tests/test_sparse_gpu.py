"""SparseEncoder and its sparse convolutions on the GPU against the dense-grid
oracle (oracle/sparse_conv_ref.py: published spconv / mmdet3d semantics; unpinned by the reference)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _cloud(rs, B, shape, n):
    D, H, W = shape
    flat = rs.choice(B * D * H * W, size=n, replace=False)
    b, r = np.divmod(flat, D * H * W)
    z, r = np.divmod(r, H * W)
    y, x = np.divmod(r, W)
    return torch.from_numpy(np.stack([b, z, y, x], 1).astype(np.int32))


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('case', ['subm', 'strided', 'strided_asym'])
def test_sparse_conv_forward_backward_vs_dense_oracle(case, dtype, tol):
    from oracle import sparse_conv_ref as R
    from unibev_amd.modules.sparse_encoder import SparseConv3d, SparseConvTensor, SubMConv3d
    rs = np.random.RandomState(4)
    B, shape, cin, cout = 2, (9, 14, 12), 16, 48
    coors = _cloud(rs, B, shape, 700)
    feats = torch.from_numpy(rs.standard_normal((700, cin)).astype(np.float32))
    if case == 'subm':
        conv = SubMConv3d(cin, cout, 3, bias=False)
        kw = None
    elif case == 'strided':
        conv = SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False)
        kw = dict(stride=(2, 2, 2), padding=(1, 1, 1))
    else:
        conv = SparseConv3d(cin, cout, (3, 1, 1), stride=(2, 1, 1), padding=0, bias=False)
        kw = dict(stride=(2, 1, 1), padding=(0, 0, 0))
    w = conv.weight.detach().clone()
    if dtype != torch.float32:                        # exactly representable operands
        feats = feats.to(dtype).float()
        w = w.to(dtype).float()
        conv.weight.data.copy_(w)
    # oracle (f64, dense grid)
    f64 = feats.double().requires_grad_()
    w64 = w.double().requires_grad_()
    dense, mask = R.densify(f64, coors, B, shape)
    ref, rmask = (R.subm_conv(dense, mask, w64) if kw is None else R.sparse_conv(dense, mask, w64, **kw))
    cot = torch.from_numpy(rs.standard_normal(tuple(ref.shape)))
    (ref * cot).sum().backward()
    # product
    conv = conv.to(DEV)
    fg = feats.to(DEV, dtype).requires_grad_()
    with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
        out = conv(SparseConvTensor(fg, coors.to(DEV), shape, B))
    assert out.features.dtype == dtype
    got = out.dense().float().cpu()
    assert got.shape == ref.shape
    act = torch.zeros(ref.shape[0], 1, *ref.shape[2:])
    c = out.indices.long().cpu()
    act[c[:, 0], 0, c[:, 1], c[:, 2], c[:, 3]] = 1
    assert torch.equal(act, rmask.float()), 'active output sites'
    scale = float(ref.detach().abs().max())
    assert float((got - ref.detach().float()).abs().max()) < tol * scale
    (out.dense().float() * cot.to(DEV).float()).sum().backward()
    gs = float(f64.grad.abs().max())
    assert float((fg.grad.float().cpu() - f64.grad.float()).abs().max()) < (tol if dtype == torch.float32 else 3e-2) * gs
    ws = float(w64.grad.abs().max())
    assert float((conv.weight.grad.cpu() - w64.grad.float()).abs().max()) < (2e-4 if dtype == torch.float32 else 3e-2) * ws


def test_sparse_conv_is_repeatable_and_order_free():
    """The rulebook holds no atomically ordered state: two runs are bit-identical, and permuting the input
    voxels permutes nothing in the dense result."""
    from unibev_amd.modules.sparse_encoder import SparseConv3d, SparseConvTensor
    rs = np.random.RandomState(9)
    B, shape = 1, (11, 20, 20)
    coors = _cloud(rs, B, shape, 1500)
    feats = torch.from_numpy(rs.standard_normal((1500, 32)).astype(np.float32))
    conv = SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False).to(DEV)
    outs = []
    for perm in (torch.arange(1500), torch.from_numpy(rs.permutation(1500)), torch.arange(1500)):
        t = SparseConvTensor(feats[perm].to(DEV), coors[perm].to(DEV), shape, B)
        outs.append(conv(t).dense())
    assert torch.equal(outs[0], outs[2])
    torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=0)


@pytest.mark.parametrize('block_type', ['basicblock', 'conv_module'])
def test_sparse_encoder_vs_dense_oracle(block_type):
    """The whole middle encoder (the reference config's structure at a small grid) in training mode: output,
    input gradient and every parameter gradient against the dense-grid oracle."""
    from oracle import sparse_conv_ref as R
    from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
    rs = np.random.RandomState(1)
    torch.manual_seed(3)
    if block_type == 'basicblock':
        cfg = dict(in_channels=5, sparse_shape=[41, 32, 32], output_channels=32, order=('conv', 'norm', 'act'),
                   encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 64), (64, 64)),
                   encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
    else:
        cfg = dict(in_channels=5, sparse_shape=[41, 32, 32], output_channels=32, order=('conv', 'norm', 'act'),
                   encoder_channels=((16,), (32, 32), (64, 64), (64, 64)),
                   encoder_paddings=((1,), (1, 1), (1, 1), ([0, 1, 1], 1)), block_type='conv_module')
    enc = build_from_cfg(dict(type='SparseEncoder', **cfg), MIDDLE_ENCODERS).to(DEV).train()
    B, n = 2, 2500
    coors = _cloud(rs, B, cfg['sparse_shape'], n)
    feats = torch.from_numpy(rs.standard_normal((n, 5)).astype(np.float32))
    fg = feats.to(DEV).requires_grad_()
    out = enc(fg, coors.to(DEV), B)
    P = {k: v.detach().cpu().double().requires_grad_() for k, v in enc.state_dict().items() if v.dtype.is_floating_point
         and 'running' not in k}
    f64 = feats.double().requires_grad_()
    ref = R.sparse_encoder(P, cfg, f64, coors, B)
    assert out.shape == ref.shape == (B, 32 * 2, 4, 4)
    cot = torch.from_numpy(rs.standard_normal(tuple(ref.shape)))
    scale = float(ref.detach().abs().max())
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 2e-4 * scale
    (ref * cot).sum().backward()
    (out.double() * cot.to(DEV)).sum().backward()
    # gradients: normwise.  A pre-activation within round-off of zero flips its ReLU between the split-bf16
    # product (2e-4 of the output scale after ~20 layers) and the f64 oracle: ~2e-4 of the units do, each
    # changing its own gradient path by 100 % and, through the batch statistics, every row a little —
    # sqrt(2e-4) ~ 1.4 % in norm.  The convolutions' own backward is held to 2e-5 / 2e-4 in the test above.
    def rel(a, b):
        return float((a.cpu().double() - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(fg.grad, f64.grad) < 3e-2
    for k, p in enc.named_parameters():
        g = P[k].grad
        assert p.grad is not None and g is not None, k
        assert rel(p.grad, g) < 3e-2, (k, rel(p.grad, g))


def test_lidar_front_end_end_to_end_at_the_reference_size():
    """points -> ubv_hard_voxelize -> VFE mean -> SparseEncoder -> (B, 256, 180, 180): the LiDAR branch up to
    the BEV feature map the point-cloud encoder consumes, at the shipped config's grid (41 x 1440 x 1440).
    Shape, finiteness, run-to-run identity, and the active columns of the output against the voxels' own
    (a stride-8 BEV cell can only be non-zero if a voxel lies within its receptive field)."""
    from unibev_amd import functional as UF
    from unibev_amd import synthetic as syn
    from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
    pts = torch.from_numpy(syn.lidar_points(30000, seed=2)).to(DEV)
    voxels, coors, num, vnum = UF.hard_voxelize(pts, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
    m = int(vnum.item())
    feats = UF.voxel_mean(voxels, num, vnum)[:m]
    c = coors[:m]
    c = torch.cat((torch.zeros_like(c[:, :1]), c[:, -3:]), 1) if c.shape[1] == 3 else c
    cfg = dict(type='SparseEncoder', in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
               order=('conv', 'norm', 'act'),
               encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
               encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
    torch.manual_seed(0)
    enc = build_from_cfg(cfg, MIDDLE_ENCODERS).to(DEV).eval()
    with torch.no_grad():
        a = enc(feats, c, 1)
        b = enc(feats, c, 1)
    assert a.shape == (1, 256, 180, 180) and torch.isfinite(a).all()
    assert torch.equal(a, b)
    # receptive field of a stride-8 cell: three stride-2 3x3 convs with padding 1 + the 3x3 submanifold ones
    yx = (c[:, 2:].long() // 8)
    occ = torch.zeros(180, 180, dtype=torch.bool, device=DEV)
    occ[yx[:, 0].clamp(0, 179), yx[:, 1].clamp(0, 179)] = True
    grown = torch.nn.functional.max_pool2d(occ[None, None].float(), 5, 1, 2)[0, 0] > 0
    nz = (a[0].abs().sum(0) > 0)
    assert nz.any() and not (nz & ~grown).any()


def test_extract_pts_feat_connects_voxelization_vfe_and_middle_encoder():
    """``extract_pts_feat`` (unibev_detector.py:111-123) over the three built pieces, two clouds of different
    sizes: equals running each stage by hand, sample by sample."""
    from unibev_amd import synthetic as syn
    from unibev_amd.modules import HardSimpleVFE, Voxelization, extract_pts_feat
    from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
    clouds = [torch.from_numpy(syn.lidar_points(n, seed=s)).to(DEV) for n, s in ((20000, 1), (12000, 2))]
    vox = Voxelization(syn.VOXEL_SIZE, syn.PC_RANGE, 10, (90000, 120000)).eval()
    vfe = HardSimpleVFE(num_features=5)
    torch.manual_seed(0)
    enc = build_from_cfg(dict(type='SparseEncoder', in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
                              order=('conv', 'norm', 'act'),
                              encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                              encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
                              block_type='basicblock'), MIDDLE_ENCODERS).to(DEV).eval()
    with torch.no_grad():
        both = extract_pts_feat(clouds, vox, vfe, enc)
        singles = [extract_pts_feat([c], vox, vfe, enc) for c in clouds]
    assert both.shape == (2, 256, 180, 180)
    for b in range(2):      # eval-mode BatchNorm: samples do not interact
        torch.testing.assert_close(both[b], singles[b][0], rtol=1e-5, atol=1e-6)


def test_rulebooks_are_kept_while_the_same_coordinates_come_back():
    """SparseEncoder keeps hash tables / neighbour maps / compacted pairs across calls for an unmodified coordinate
    tensor (identity + version): same results bit for bit, no rebuild; an in-place edit or a new tensor rebuilds."""
    from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
    rs = np.random.RandomState(3)
    cfg = dict(type='SparseEncoder', in_channels=5, sparse_shape=[41, 32, 32], output_channels=32,
               encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 64), (64, 64)),
               encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
    torch.manual_seed(0)
    enc = build_from_cfg(cfg, MIDDLE_ENCODERS).to(DEV).train()
    assert enc.keep_rulebooks is False            # opt-in: the cache trusts the tensor's identity + version counter
    enc.keep_rulebooks = True
    coors = _cloud(rs, 2, cfg['sparse_shape'], 2000).to(DEV)
    feats = torch.from_numpy(rs.standard_normal((2000, 5)).astype(np.float32)).to(DEV).requires_grad_()
    outs, grads = [], []
    for _ in range(2):
        for p in enc.parameters():
            p.grad = None
        y = enc(feats, coors, 2)
        y.square().sum().backward()
        outs.append(y.detach().clone())
        grads.append(enc.conv_input[0].weight.grad.clone())
    book = next(iter(enc._rulebooks.values()))[3]
    assert torch.equal(outs[0], outs[1]) and torch.equal(grads[0], grads[1])
    assert 'subm1' in book and book['subm1'][4].get('pairs') is not None       # compacted pairs built by the backward
    same = enc._rulebooks[id(coors)][3]
    enc(feats, coors, 2)
    assert enc._rulebooks[id(coors)][3] is same                                # reused
    coors[0, 3] = (coors[0, 3] + 1) % 32                                       # in-place edit: version moves
    enc(feats, coors, 2)
    assert enc._rulebooks[id(coors)][3] is not same
    fresh = build_from_cfg(cfg, MIDDLE_ENCODERS).to(DEV).train()
    fresh.load_state_dict(enc.state_dict())
    torch.testing.assert_close(enc(feats, coors, 2), fresh(feats, coors.clone(), 2), rtol=0, atol=0)
    assert not fresh._rulebooks                   # the default keeps nothing


@pytest.mark.parametrize('C,relu', [(16, True), (32, False), (64, True), (128, True)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_rows_batch_norm_vs_torch(C, relu, dtype):
    """Fused BatchNorm1d (+ ReLU) over sparse feature rows: output, running statistics, and the gradients of x / gamma /
    beta equal torch.nn.BatchNorm1d (+ ReLU) in f64 on the same values — training and eval mode."""
    from unibev_amd.functional import rows_batch_norm
    rs = np.random.RandomState(C)
    N = 5000 + C
    x = torch.from_numpy(rs.standard_normal((N, C)).astype(np.float32) * 2 + 0.5).to(dtype)
    cot = torch.from_numpy(rs.standard_normal((N, C)).astype(np.float32)).to(dtype)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for training in (True, False):
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(DEV).train(training)
        ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).double().train(training)
        with torch.no_grad():
            for m in (bn, ref):
                m.weight.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, C)))
                m.bias.copy_(torch.from_numpy(rs.standard_normal(C) * 0.3))
                m.running_mean.copy_(torch.from_numpy(rs.standard_normal(C) * 0.2))
                m.running_var.copy_(torch.from_numpy(rs.uniform(0.5, 2.0, C)))
            ref.weight.copy_(bn.weight.double().cpu()); ref.bias.copy_(bn.bias.double().cpu())
            ref.running_mean.copy_(bn.running_mean.double().cpu()); ref.running_var.copy_(bn.running_var.double().cpu())
        xg = x.to(DEV).requires_grad_()
        y = rows_batch_norm(xg, bn, relu=relu)
        assert y is not None and y.dtype == dtype
        x64 = x.double().requires_grad_()
        r = ref(x64)
        r = torch.relu(r) if relu else r
        (y.float() * cot.to(DEV).float()).sum().backward()
        (r * cot.double()).sum().backward()
        rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
        assert rel(y, r.detach()) < tol
        assert rel(xg.grad, x64.grad) < (tol if dtype == torch.float32 else 4e-2)
        assert rel(bn.weight.grad, ref.weight.grad) < tol and rel(bn.bias.grad, ref.bias.grad) < tol
        assert rel(bn.running_mean, ref.running_mean) < 1e-5 and rel(bn.running_var, ref.running_var) < 1e-5
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


def test_rows_batch_norm_large_mean_and_module_edge_cases():
    """Statistics are sums of x - x[0]: a channel whose mean is 1000x its spread keeps its variance (E[x^2] - mean^2 of the
    raw f32 values cancels there); one row in training mode and non-f32 running statistics are left to the module."""
    from unibev_amd.functional import rows_batch_norm
    rs = np.random.RandomState(0)
    N, C = 20000, 32
    x = torch.from_numpy((rs.standard_normal((N, C)) * 0.5 + 500.0).astype(np.float32))
    bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(DEV).train()
    y = rows_batch_norm(x.to(DEV), bn)
    ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).double().train()(x.double())
    assert float((y.double().cpu() - ref).abs().max()) < 2e-3            # (the inputs' own f32 spacing at 500 is 3e-5)
    v = x.double().var(0, unbiased=True)
    assert float(((bn.running_var.double().cpu() - 0.99) / 0.01 - v).abs().max() / v.max()) < 1e-3
    assert rows_batch_norm(x[:1].to(DEV), bn) is None                    # torch raises for one value per channel
    half = torch.nn.BatchNorm1d(C).to(DEV).half().train()
    assert rows_batch_norm(x.to(DEV), half) is None


def _np_sites(coors, B, in_dims, ksize, stride, pad):
    """Reference output set of a strided sparse convolution (numpy): ascending keys -> coordinates."""
    out_dims = tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip(in_dims, ksize, stride, pad))
    keys = set()
    c = coors.numpy()
    for kz in range(ksize[0]):
        for ky in range(ksize[1]):
            for kx in range(ksize[2]):
                num = c[:, 1:] + np.array(pad) - np.array([kz, ky, kx])
                ok = (num >= 0).all(1) & (num % np.array(stride) == 0).all(1)
                t = num // np.array(stride)
                ok &= (t < np.array(out_dims)).all(1)
                k = ((c[:, 0].astype(np.int64) * out_dims[0] + t[:, 0]) * out_dims[1] + t[:, 1]) * out_dims[2] + t[:, 2]
                keys.update(k[ok].tolist())
    keys = np.array(sorted(keys), dtype=np.int64)
    x = keys % out_dims[2]; r = keys // out_dims[2]
    y = r % out_dims[1]; r //= out_dims[1]
    z = r % out_dims[0]; b = r // out_dims[0]
    return np.stack([b, z, y, x], 1).astype(np.int32), out_dims


def test_output_sites_chain_matches_the_layer_by_layer_sets():
    """ubv_spconv_output_sites: bit-exact against a host enumeration, for a chain of three strided layers built with
    one host read (device-side input counts) and for an empty input."""
    from unibev_amd import functional as UF
    rs = np.random.RandomState(11)
    B, shape = 2, (17, 40, 36)
    coors = _cloud(rs, B, shape, 3000)
    layers = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
    chain = UF.spconv_site_chain(coors.to(DEV), B, shape, layers)
    cur, dims = coors, shape
    for (oc, od), (k, s, p) in zip(chain, layers):
        ref, rd = _np_sites(cur, B, dims, k, s, p)
        assert tuple(od) == tuple(rd)
        assert oc.shape == ref.shape and np.array_equal(oc.cpu().numpy(), ref)
        cur, dims = torch.from_numpy(ref), rd
    # a stride-1 SparseConv3d dilates the set (27 candidate outputs per input)
    (oc, od), = UF.spconv_site_chain(coors.to(DEV), B, shape, [((3, 3, 3), (1, 1, 1), (1, 1, 1))])
    ref, rd = _np_sites(coors, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert np.array_equal(oc.cpu().numpy(), ref)
    # no input rows: no output rows
    (oc, od), = UF.spconv_site_chain(coors[:0].to(DEV), B, shape, layers[:1])
    assert oc.shape == (0, 4)


def test_compacted_pairs_match_a_stable_sort_of_the_neighbour_map():
    from unibev_amd import functional as UF
    rs = np.random.RandomState(12)
    for rows in (1, 63, 2048, 2049, 7001):
        nbr = torch.from_numpy(rs.randint(-3, rows, size=(27, rows)).astype(np.int32)).clamp_(min=-1)
        nbr[5] = -1                                           # an offset without any pair
        nbr[6] = torch.arange(rows, dtype=torch.int32)        # a full one
        out_rows, in_rows, counts = UF.spconv_pairs(nbr.to(DEV))
        out_rows, in_rows, counts = out_rows.cpu(), in_rows.cpu(), counts.cpu()
        for k in range(27):
            valid = torch.nonzero(nbr[k] >= 0)[:, 0]
            n = int(counts[k])
            assert n == valid.numel()
            assert torch.equal(out_rows[k, :n].long(), valid)
            assert torch.equal(in_rows[k, :n], nbr[k][valid])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('relu', [False, True])
def test_rows_batch_norm_with_residual_vs_torch(relu, dtype):
    """The basic block's tail relu?(bn(x) + identity) as one pass each way: output and the gradients of x, the identity,
    gamma and beta equal torch in f64; training and eval mode; counters under ``batched_bn_ticks``."""
    from unibev_amd.functional import batched_bn_ticks, rows_batch_norm
    rs = np.random.RandomState(7)
    N, C = 4111, 64
    x = torch.from_numpy(rs.standard_normal((N, C)).astype(np.float32) * 2 + 0.5).to(dtype)
    idn = torch.from_numpy(rs.standard_normal((N, C)).astype(np.float32)).to(dtype)
    cot = torch.from_numpy(rs.standard_normal((N, C)).astype(np.float32)).to(dtype)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for training in (True, False):
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(DEV).train(training)
        ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).double().train(training)
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, C)))
            bn.bias.copy_(torch.from_numpy(rs.standard_normal(C) * 0.3))
            bn.running_var.copy_(torch.from_numpy(rs.uniform(0.5, 2.0, C)))
            ref.weight.copy_(bn.weight.double().cpu()); ref.bias.copy_(bn.bias.double().cpu())
            ref.running_var.copy_(bn.running_var.double().cpu())
        xg, ig = x.to(DEV).requires_grad_(), idn.to(DEV).requires_grad_()
        with batched_bn_ticks():
            y = rows_batch_norm(xg, bn, relu=relu, residual=ig)
            assert int(bn.num_batches_tracked) == 0                    # (deferred to the context's exit)
        assert int(bn.num_batches_tracked) == (1 if training else 0)
        assert y is not None and y.dtype == dtype
        x64, i64 = x.double().requires_grad_(), idn.double().requires_grad_()
        r = ref(x64) + i64
        r = torch.relu(r) if relu else r
        (y.float() * cot.to(DEV).float()).sum().backward()
        (r * cot.double()).sum().backward()
        rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
        assert rel(y, r.detach()) < tol
        gt = tol if dtype == torch.float32 else 4e-2
        assert rel(xg.grad, x64.grad) < gt and rel(ig.grad, i64.grad) < gt
        assert rel(bn.weight.grad, ref.weight.grad) < tol and rel(bn.bias.grad, ref.bias.grad) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_weight_operand_kernel_equals_the_framework_composition(dtype):
    """ubv_spconv_weight_operand (transpose / mirror / pad / split in one launch) is bit-identical to
    reshape + transpose / flip + pad + split."""
    from unibev_amd import functional as UF
    rs = np.random.RandomState(3)
    for cin, cout in ((16, 48), (64, 128), (32, 16)):
        w = torch.from_numpy(rs.standard_normal((27, cin, cout)).astype(np.float32)).to(DEV, dtype)
        for transpose, flip in ((True, False), (False, False), (False, True)):
            hi, lo = UF.spconv_weight_operand(w, transpose, flip)
            src = w.transpose(1, 2) if transpose else (w.flip(0) if flip else w)
            rhi, rlo = UF.spconv_operand(src)
            assert torch.equal(hi.view(torch.int16).flatten(), rhi.view(torch.int16).flatten())
            if dtype == torch.float32:
                assert torch.equal(lo.view(torch.int16).flatten(), rlo.view(torch.int16).flatten())
            else:
                assert lo is None and rlo is None


@pytest.mark.parametrize('pw', [4, 8])
def test_sparse_weight_gradient_on_the_wave_specialised_kernel(pw):
    """ubv_spconv_wgrad_pairs through csrc/gemm_wgrad_ws.inl (pair indices fetched two chunks ahead of their rows, offsets
    packed into the 128-wide tile as in the 4-wave kernel, which is the default): the convolution and encoder tests
    above, re-run with that kernel selected (ubv_debug_set_wgrad_ws)."""
    from unibev_amd._lib import lib
    assert lib().ubv_debug_set_wgrad_ws(-1, pw) == 0
    try:
        for case in ('subm', 'strided', 'strided_asym'):
            test_sparse_conv_forward_backward_vs_dense_oracle(case, torch.float32, 2e-5)
        test_sparse_conv_forward_backward_vs_dense_oracle('subm', torch.bfloat16, 2e-2)
        test_sparse_encoder_vs_dense_oracle('basicblock')
    finally:
        lib().ubv_debug_set_wgrad_ws(-1, 0)
