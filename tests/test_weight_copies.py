"""bf16 working copies of the fp32 Linear weights (vitadapter/fused.py::_Bf16Copies): a parameter
written through `.data` (which does not bump Tensor._version: legacy optimizers, EMA hooks,
`weight.data.normal_()` in the reference's _init_weights, ref segmentation/.../vit_adapter.py:61-74) must
be seen by the next forward, and a copy saved for a backward must never change under it."""
import pytest
import torch

from vitadapter import fused


def test_copies_follow_data_writes_cpu():
    cp = fused._Bf16Copies()
    p = torch.nn.Parameter(torch.arange(8, dtype=torch.float32))
    # outside a forward epoch every use casts afresh
    a = cp.get(p)
    p.data.mul_(2)                       # _version unchanged by construction of .data
    b = cp.get(p)
    assert torch.equal(b.float(), p.detach()) and not torch.equal(a, b)
    # inside an epoch: one copy serves every use, made from the current values
    cp.begin([p])
    c1, c2 = cp.get(p), cp.get(p)
    assert c1 is c2 and torch.equal(c1.float(), p.detach())
    cp.end()
    p.data.add_(1)
    cp.begin([p])
    c3 = cp.get(p)
    assert torch.equal(c3.float(), p.detach())
    assert torch.equal(c1.float(), p.detach() - 1), 'a copy handed out earlier must not be rewritten in place'
    cp.end()
    # after the epoch closes the stale epoch copy is not served
    p.data.add_(1)
    assert torch.equal(cp.get(p).float(), p.detach())


@pytest.mark.gpu
def test_fused_linear_sees_data_writes_gpu():
    lin = torch.nn.Linear(64, 32).cuda()
    pair_a, pair_b = torch.nn.Linear(64, 16).cuda(), torch.nn.Linear(64, 8).cuda()
    x = torch.randn(128, 64, device='cuda')
    holder = torch.nn.ModuleList([lin, pair_a, pair_b])

    def run():
        with torch.autocast('cuda', dtype=torch.bfloat16), fused.forward_epoch(holder):
            y = fused.linear(lin, x)
            ya, yb = fused.linear_pair(pair_a, pair_b, x)
        return y.float(), ya.float(), yb.float()

    y0, a0, b0 = run()
    for m in (lin, pair_a, pair_b):
        m.weight.data.mul_(2)
        m.bias.data.zero_()
    y1, a1, b1 = run()
    for got, m in ((y1, lin), (a1, pair_a), (b1, pair_b)):
        want = torch.nn.functional.linear(x, m.weight, m.bias)
        assert (got - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
    assert not torch.allclose(y0, y1)


@pytest.mark.gpu
def test_backward_uses_the_copy_of_its_own_forward_gpu():
    """forward A, parameters change, forward B, then backward A: dX of A must use A's weights."""
    lin = torch.nn.Linear(64, 32).cuda()
    holder = torch.nn.ModuleList([lin])
    x = torch.randn(16, 64, device='cuda', requires_grad=True)
    w0 = lin.weight.detach().clone()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        with fused.forward_epoch(holder):
            ya = fused.linear(lin, x)
        lin.weight.data.mul_(3)
        with fused.forward_epoch(holder):
            yb = fused.linear(lin, x.detach())
    g = torch.randn_like(ya)
    (gx,) = torch.autograd.grad(ya, x, g)
    want = g.float() @ w0.to(torch.bfloat16).float()
    assert (gx.float() - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
    assert yb is not None


def test_side_stream_deferral_rules():
    """A weight gradient may stay on the side stream (and be stored by the join instead of by autograd) only when this
    backward pass does nothing with it but accumulate it into .grad (vitadapter/fused.py::_SideStream): leaf parameters
    of an open forward epoch, no tensor hooks, AccumulateGrad scheduled - not under autograd.grad(), not when
    backward(inputs=...) leaves the parameter out."""
    side = fused._SideStream()
    w = torch.nn.Parameter(torch.zeros(4))
    b = torch.nn.Parameter(torch.zeros(4))
    seen = {}

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.tok = side.note(w, b)
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            seen['tok'] = ctx.tok
            seen['ok'] = ctx.tok is not None and all(side._only_accumulated(r()) for r in ctx.tok)
            return g, torch.zeros_like(w), torch.zeros_like(b)

    x = torch.zeros(4, requires_grad=True)
    assert side.may_defer(None, torch.device('cpu')) is False
    Probe.apply(x, w, b).sum().backward()
    assert seen['tok'] is None                                        # no forward epoch open: working copies are per use
    fused.BF16_COPIES.epoch = 1
    try:
        assert side.note(w.view(4), None) is None                     # not a leaf
        Probe.apply(x, w, b).sum().backward()
        assert seen['ok'] is True and seen['tok'][0]() is w and seen['tok'][1]() is b
        assert side.may_defer(seen['tok'], torch.device('cpu')) is False          # not a GPU tensor
        torch.autograd.grad(Probe.apply(x, w, b).sum(), [w])          # gradients are returned, not accumulated
        assert seen['ok'] is False
        Probe.apply(x, w, b).sum().backward(inputs=[x])               # the parameters are left out of this pass
        assert seen['ok'] is False
        Probe.apply(x, w, b).sum().backward(inputs=[w, b])
        assert seen['ok'] is True
        h = w.register_hook(lambda g: g)                              # somebody looks at the gradient in mid-pass
        Probe.apply(x, w, b).sum().backward()
        assert seen['ok'] is False
        h.remove()
        Probe.apply(x, w, b).sum().backward()
        assert seen['ok'] is True
        h = b.register_post_accumulate_grad_hook(lambda p: None)      # e.g. an optimizer step inside the backward
        Probe.apply(x, w, b).sum().backward()
        assert seen['ok'] is False
        h.remove()
    finally:
        fused.BF16_COPIES.epoch = 0


def _lin_grads(lin, holder, xs, retain=False, extra_use=False):
    """Gradients of a fused Linear after (a) several forwards and ONE backward, (b) a second backward on a retained
    graph, (c) a second use of the weight by a plain operator in the same graph."""
    lin.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        total = 0
        for x in xs:
            with fused.forward_epoch(holder):
                y = fused.linear(lin, x)
            total = total + (y.float() ** 2).sum()
            if extra_use:
                total = total + torch.nn.functional.linear(x.float(), lin.weight.float()).sum()
    total.backward(retain_graph=retain)
    if retain:
        total.backward()
    torch.cuda.synchronize()
    return lin.weight.grad.clone(), lin.bias.grad.clone()


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['two_forwards_one_backward', 'retain_graph_double_backward', 'weight_used_by_another_op'])
def test_side_stream_gradients_equal_main_stream_gradients(case):
    """the weight-gradient GEMMs left on the side stream must never race with autograd's accumulation.  Same
    graph with the overlap on and off (VAH_FUSED_DISABLE=wgrad_overlap): identical gradients, repeated so that a race
    would have room to show."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(768, 768).cuda()
    holder = torch.nn.ModuleList([lin])
    n = 2 if case == 'two_forwards_one_backward' else 1
    xs = [torch.randn(8192, 768, device='cuda') for _ in range(n)]
    kw = dict(retain=case == 'retain_graph_double_backward', extra_use=case == 'weight_used_by_another_op')
    old = fused.ENABLED['wgrad_overlap']
    try:
        fused.ENABLED['wgrad_overlap'] = False
        want = _lin_grads(lin, holder, xs, **kw)
        fused.ENABLED['wgrad_overlap'] = True
        for _ in range(5):
            got = _lin_grads(lin, holder, xs, **kw)
            for g, w in zip(got, want):
                assert torch.equal(g, w), case
    finally:
        fused.ENABLED['wgrad_overlap'] = old
