"""Host logic of unibev_amd.linear / functional that needs no GPU: the low-precision shadow refresh plan
and the gradient tags."""
import torch

from unibev_amd import functional as UF
from unibev_amd import linear as UL


def test_refresh_plan_follows_a_replaced_shadow():
    """the cached refresh plan was invalidated only by len(_SHADOWS) changing; a shadow replaced
    under its key kept being refreshed into the OLD buffer."""
    saved = dict(UL._SHADOWS), UL._ACTIVE
    try:
        UL._SHADOWS.clear()
        UL._ACTIVE = True
        w = torch.nn.Parameter(torch.randn(8, 4))
        b1 = UL._cached_lowp([w], torch.bfloat16)
        plan = UL._refresh_plan()
        assert any(v.data_ptr() == b1.data_ptr() for g in plan['groups'] for v in g[0])
        # replace the shadow under the same key (what the device-mismatch path does)
        key = next(iter(UL._SHADOWS))
        del UL._SHADOWS[key]
        UL._PLAN['n'] = 1                       # the stale state the finding describes: same length as before
        b2 = UL._cached_lowp([w], torch.bfloat16)
        assert b2.data_ptr() != b1.data_ptr() and len(UL._SHADOWS) == 1
        plan = UL._refresh_plan()
        ptrs = [v.data_ptr() for g in plan['groups'] for v in g[0]]
        assert b2.data_ptr() in ptrs and b1.data_ptr() not in ptrs
    finally:
        UL._SHADOWS.clear()
        UL._SHADOWS.update(saved[0])
        UL._ACTIVE = saved[1]
        UL._PLAN['n'] = -1


def test_gradient_tags_go_stale_when_autograd_accumulates_in_place():
    """a tag on a gradient tensor must not survive a second consumer's in-place accumulation."""
    g = torch.zeros(4, 3)
    UF.tag_grad(g, '_ubv_colsum', torch.ones(3))
    assert UF.grad_tag(g, '_ubv_colsum') is not None and not UF.grad_tag_stale(g, '_ubv_colsum')
    g.add_(1.0)                                  # what autograd's InputBuffer does for a second consumer
    assert UF.grad_tag(g, '_ubv_colsum') is None and UF.grad_tag_stale(g, '_ubv_colsum')
    assert UF.grad_tag(torch.zeros(2), '_ubv_colsum') is None


def test_second_consumer_of_a_tagged_gradient_gets_the_true_bias_gradient():
    """Linear -> (LayerNorm-like producer tagging column sums) with a second consumer of the Linear output:
    the stale column sum must not become the bias gradient."""
    lin = torch.nn.Linear(6, 5)
    x = torch.randn(7, 6)

    class Producer(torch.autograd.Function):     # tags its input gradient with (wrong once accumulated) sums
        @staticmethod
        def forward(ctx, t):
            return t * 2.0

        @staticmethod
        def backward(ctx, g):
            gx = (g * 2.0).contiguous()
            UF.tag_grad(gx, '_ubv_colsum', gx.sum(0))
            return gx

    y = UL.linear(x, lin.weight, lin.bias)
    out = Producer.apply(y).sum() + (y * 3.0).sum()          # second consumer of y
    out.backward()
    ref = torch.nn.functional.linear(x, lin.weight.detach(), lin.bias.detach().requires_grad_())
    want = torch.full((5,), 7 * 5.0)                         # d/db of sum(2 y) + sum(3 y) over 7 rows
    torch.testing.assert_close(lin.bias.grad, want)
