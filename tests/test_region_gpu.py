"""No framework kernel inside the two-stream windows of the encoders.

The image and point-cloud encoders run on two HIP streams (modules/transformer.py) — forward, and again when autograd
replays the backward on the forward ops' streams.  Every kernel of this library is plain f32 (csrc/Makefile,
tests/test_build_isa.py) because kernels with packed f32 instructions returned wrong results beside another stream's
MFMA + VALU kernels (profiles/r05_pk_mfma_hazard.txt); the framework's element-wise kernels carry no such guarantee.
``unibev_amd.debug.ForeignKernelLog`` records every aten operator that touches device memory between a fork and its
join, and every C-ABI launch; the training step of the bench's workload must show none of the former."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (Until the round's last session the engine still summed the two encoders' gradients of the BEV query table and the
# positional table with a framework add where the branches meet; both tables now part in functional.fan_out ahead of the
# fork and their gradients meet in this library's add.)  NOTHING is allowed.
JOIN_ADDS = set()


def _step_log(dtype=torch.float32):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B
    from unibev_amd.debug import ForeignKernelLog
    from unibev_amd.modules import transformer as TR
    dev = torch.device('cuda', 0)
    prev = torch.cuda.current_stream(dev)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    try:
        torch.manual_seed(0)
        head, _ = B.build_head('LC_cnw', dev)
        head.train()
        img, pts, metas = B.synth_inputs('LC_cnw', 2, dtype, dev, 0)
        params = [p for p in head.parameters() if p.requires_grad]
        cot = torch.randn(200 * 200, 2, 256, device=dev) / 200.0
        head.transformer.forced_flags = (1, 1)

        def step(log=None):
            for p in params + img + pts:
                p.grad = None
            out = head.forward_bev(img, pts, metas)
            if log is not None:
                log.mark('backward')
            out.backward(cot)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        with ForeignKernelLog(TR._side_stream(dev)) as log:
            step(log)
        torch.cuda.synchronize()
        return log
    finally:
        torch.cuda.set_stream(prev)


def test_no_framework_kernel_between_fork_and_join():
    log = _step_log()
    summary = log.summary()
    # both windows were seen, both streams were busy in them, with this library's launches
    assert set(summary) == {'forward', 'backward'}, summary
    assert summary['forward']['own'] >= 60 and summary['forward']['side'] >= 25, summary
    assert summary['backward']['own'] >= 100 and summary['backward']['side'] >= 50, summary
    bad = {k: v for k, v in log.offenders().items() if k not in JOIN_ADDS}
    assert not bad, 'framework operators inside the two-stream window:\n' + '\n'.join(f'{v} x {k}' for k, v in bad.items())
    assert sum(log.offenders().values()) == 0, log.offenders()


def test_own_add_matches_torch():
    from unibev_amd import functional as UF
    g = torch.Generator(device='cpu').manual_seed(5)
    for n in (4, 64 * 256, 40000 * 96, 1027):
        a = torch.randn(n, generator=g).cuda()
        b = torch.randn(n, generator=g).cuda()
        assert torch.equal(UF.add2(a, b), a + b)
    a = torch.randn(96, 256, generator=g).cuda()
    b = torch.randn(96, 256, generator=g).cuda()
    want = a + b
    assert UF.add2(a, b, out=a) is a and torch.equal(a, want)


def test_fan_out_sums_the_two_gradients_with_the_own_kernel():
    from unibev_amd import functional as UF
    w = torch.nn.Parameter(torch.randn(64, 256, device='cuda'))
    a, b = UF.fan_out(w)
    assert a._ubv_master is w and b._ubv_master is w and a.data_ptr() == w.data_ptr()
    ((a * 2.0).sum() + (b * 3.0).sum()).backward()
    assert torch.equal(w.grad, torch.full_like(w, 5.0))
    w.grad = None
    a, b = UF.fan_out(w)
    (a * 2.0).sum().backward()                   # one consumer only: its gradient passes through
    assert torch.equal(w.grad, torch.full_like(w, 2.0))


def test_fan_out_pair_adds_adjacent_slices_once():
    from unibev_amd import functional as UF
    w1 = torch.nn.Parameter(torch.randn(64, 256, device='cuda'))
    w2 = torch.nn.Parameter(torch.randn(32, 256, device='cuda'))
    (a1, b1), (a2, b2) = UF.fan_out_pair(w1, w2)
    ga = torch.randn(96, 256, device='cuda')
    gb = torch.randn(96, 256, device='cuda')
    torch.autograd.backward([a1, a2, b1, b2], [ga[:64], ga[64:], gb[:64], gb[64:]])      # adjacent slices: one add
    assert torch.equal(w1.grad, (ga + gb)[:64]) and torch.equal(w2.grad, (ga + gb)[64:])
    w1.grad = w2.grad = None
    (a1, b1), (a2, b2) = UF.fan_out_pair(w1, w2)
    torch.autograd.backward([a1, a2, b1, b2], [ga[:64].clone(), ga[64:].clone(), gb[:64].clone(), gb[64:].clone()])
    assert torch.equal(w1.grad, (ga + gb)[:64]) and torch.equal(w2.grad, (ga + gb)[64:])


def test_slice_sum_writes_a_column_slice():
    """``ubv_slice_sum_f32``: the batch sum of S row blocks written into a column slice of a wider matrix (the positional
    fold's gradient slot); bit-equal to the same sum in the same order."""
    from unibev_amd import functional as UF
    g = torch.Generator(device='cpu').manual_seed(9)
    for S, rows, cols, ld, c0 in ((2, 40000, 96, 288, 96), (1, 1000, 96, 288, 192), (3, 77, 32, 32, 0)):
        x = torch.randn(S, rows, cols, generator=g).cuda()
        G = torch.full((rows, ld), 7.0, device='cuda')
        out = UF.slice_sum(x, G[:, c0:c0 + cols])
        want = x[0].clone()
        for s in range(1, S):
            want = want + x[s]
        assert out.data_ptr() == G[:, c0:c0 + cols].data_ptr()
        assert torch.equal(G[:, c0:c0 + cols], want)
        keep = torch.ones(ld, dtype=torch.bool)
        keep[c0:c0 + cols] = False
        assert bool((G[:, keep.cuda()] == 7.0).all())         # nothing outside the slice is touched


def test_pos_fold_all_matches_separate_linears():
    """``linear.pos_fold_all``: per-layer positional terms from ONE GEMM, their gradients through the slots — equal to the
    per-layer ``F.linear`` products and their autograd gradients (split-bf16 GEMM tolerance)."""
    import torch.nn.functional as F
    from unibev_amd import linear as UL
    g = torch.Generator(device='cpu').manual_seed(11)
    R, C, B = 4096, 256, 2
    base = torch.randn(R, C, generator=g).cuda().requires_grad_()
    ws = [torch.nn.Parameter(torch.randn(n, C, generator=g).cuda() / 16) for n in (64, 32, 64, 32)]
    gol = [torch.randn(B, R, 96, generator=g).cuda() for _ in range(2)]
    with UL.lowp_step_cache():
        terms = UL.pos_fold_all(base, ws)
        assert len(terms) == 2 and terms[0].shape == (R, 96) and hasattr(terms[0], '_ubv_grad_slot')
        # a consumer's backward: the batch sum of its offset | logit gradient goes to the slot
        gs = [UL._row_bias_grad(gol[i], terms[i]._ubv_grad_slot) for i in range(2)]
        torch.autograd.backward(terms, gs)
    ref_base = base.detach().clone().requires_grad_()
    ref_ws = [w.detach().clone().requires_grad_() for w in ws]
    ref_terms = [F.linear(ref_base, torch.cat((ref_ws[2 * i], ref_ws[2 * i + 1]))) for i in range(2)]
    torch.autograd.backward(ref_terms, [gol[i].sum(0) for i in range(2)])
    for i in range(2):
        assert float((terms[i] - ref_terms[i]).norm() / ref_terms[i].norm()) < 1e-4
    assert float((base.grad - ref_base.grad).norm() / ref_base.grad.norm()) < 1e-4
    for w, rw in zip(ws, ref_ws):
        assert float((w.grad - rw.grad).norm() / rw.grad.norm()) < 1e-4
