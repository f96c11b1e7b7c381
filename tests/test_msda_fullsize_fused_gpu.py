"""GPU: the fused MSDeformAttn core (what bench.py runs) at the FULL call shapes of BASELINE configs[2]
(ViT-Adapter-B 1024x1024, batch 2: injector Lq=4096 over 128^2+64^2+32^2 value rows, extractor Lq=21504
over 64^2, 12 heads x 32) against the C oracle (oracle/msda_oracle.c: the scalar statement of
/root/reference/detection/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159,237-403) run at the same size in
fp64 on the SAME bf16-rounded operands.

Checked per case: forward out, grad_value (every element: the tile pass must place every contribution of
every query), d(offsets), d(logits) - for bf16 IO with a tolerance that is the bf16 rounding of the result
itself (2^-8 relative per element + 2e-3 of the tensor's max for the elements near zero: operands are
exact bf16 on both sides, products and sums are fp32 on the GPU and fp64 in the oracle), for fp32 IO with
the north-star 1e-4.  The reference's module does the location / softmax arithmetic in PyTorch around the
kernel (/root/reference/detection/ops/modules/ms_deform_attn.py:108-128); here it is part of the oracle
side of the comparison, in fp64.
"""
import numpy as np
import pytest
import torch

from oracle import cases
from oracle import msda as oracle_msda

pytestmark = pytest.mark.gpu


def _inputs(cfg, dtype, noise, seed=0):
    N, M, D, P, Lq, shapes, qshapes = cases.bench_inputs(cfg)
    L, S = len(shapes), sum(h * w for h, w in shapes)
    g = torch.Generator(device='cuda').manual_seed(seed)
    value = torch.randn(N, S, M, D, device='cuda', generator=g).to(dtype)
    off = (cases.ring_offsets(M, L, P).cuda()[None, None]
           + noise * torch.randn(N, Lq, M, L, P, 2, device='cuda', generator=g)).to(dtype)
    logit = torch.randn(N, Lq, M, L * P, device='cuda', generator=g).to(dtype)
    gout = torch.randn(N, Lq, M * D, device='cuda', generator=g).to(dtype)
    ref = cases.reference_grid(qshapes).cuda()
    hw = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
    lsi = cases.level_start_index(shapes).cuda()
    return value, off, logit, gout, ref, hw, lsi, (N, M, D, P, Lq, L, S, shapes)


def _oracle(value, off, logit, gout, ref, shapes, dims):
    """fp64 oracle on the (bf16-rounded) operands -> out, grad_value, d_off, d_logit."""
    N, M, D, P, Lq, L, S, _ = dims
    v = value.double().cpu().numpy()
    o = off.double().cpu()
    lg = logit.double().cpu()
    r = ref.double().cpu()                                                  # (1, Lq, 1, 2)
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float64)     # (L, 2) as (W, H)
    loc = r[:, :, None, :, None, :] + o / wh[None, None, None, :, None, :]
    attn = torch.softmax(lg, -1).view(N, Lq, M, L, P)
    hw = np.asarray(shapes, dtype=np.int64)
    lsi = oracle_msda.level_start_index(shapes)
    out = oracle_msda.forward(v, hw, lsi, loc.numpy(), attn.numpy())
    gv, gl, ga = oracle_msda.backward(v, hw, lsi, loc.numpy(), attn.numpy(), gout.double().cpu().numpy())
    d_off = torch.from_numpy(gl) / wh[None, None, None, :, None, :]
    # bilinear interpolation has a kink where a pixel coordinate is an integer: d(out)/d(loc) jumps there, and
    # bf16 offsets (multiples of 2^-7 px near 1..4) land on integers often.  fp32 (GPU) and fp64 (oracle)
    # locations then fall on different sides; such samples are left out of the d(offsets) comparison (the
    # other three results are continuous in the location).
    px = loc * wh[None, None, None, :, None, :] - 0.5
    smooth = ((px - px.round()).abs() > 1e-3).all(-1, keepdim=True).expand_as(d_off).numpy()
    ga = torch.from_numpy(ga).view(N, Lq, M, L * P)
    p = attn.view(N, Lq, M, L * P)
    d_logit = p * (ga - (p * ga).sum(-1, keepdim=True))
    return out, gv, d_off.numpy(), d_logit.numpy(), smooth


def _check(name, got, want, bf16, mask=None):
    got = got.detach().double().cpu().numpy().reshape(want.shape)
    assert np.isfinite(got).all(), name
    if mask is not None:
        assert mask.mean() > 0.5, 'more than half of the samples sit on a kink'
        got, want = np.where(mask, got, 0.0), np.where(mask, want, 0.0)
    err = np.abs(got - want)
    scale = np.abs(want).max()
    if bf16:
        excess = err - 2.0 ** -8 * np.abs(want)
        assert excess.max() <= 2e-3 * scale, '%s: %.3e over the bf16 rounding band (max |ref| %.3e)' % (
            name, excess.max(), scale)
        rel_l2 = np.sqrt((err ** 2).sum() / (want ** 2).sum())
        assert rel_l2 <= 4e-3, '%s: relative L2 error %.3e' % (name, rel_l2)
    else:
        assert err.max() <= 1e-4 * max(1.0, scale), '%s: %.3e (max |ref| %.3e)' % (name, err.max(), scale)


@pytest.mark.parametrize('cfg', ['cfg3_inj', 'cfg3_ext'])
@pytest.mark.parametrize('dtype,noise', [(torch.bfloat16, 0.0), (torch.bfloat16, 1.0), (torch.float32, 1.0)])
def test_fused_core_full_size_vs_oracle(cfg, dtype, noise):
    """noise 0: the offsets of a freshly initialised model (what bench.py runs: every query of a head has the
    same offsets), noise 1: ring bias + N(0,1) px.  Both run the tile pass (ops.functions.ms_deform_attn_fused:
    tiled_backward) with lists of several 64-entry chunks per tile on the extractor shape."""
    from ops.functions import MSDeformAttnFusedFunction
    from ops.functions import ms_deform_attn_fused as mf
    value, off, logit, gout, ref, hw, lsi, dims = _inputs(cfg, dtype, noise)
    assert mf.tiled_backward(dims[5], dims[3])
    value.requires_grad_(True)
    off.requires_grad_(True)
    logit.requires_grad_(True)
    out = MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref)
    gv, d_off, d_logit = torch.autograd.grad(out, [value, off, logit], gout)
    assert out.dtype == dtype and gv.dtype == dtype and d_off.dtype == dtype
    w_out, w_gv, w_off, w_logit, smooth = _oracle(value.detach(), off.detach(), logit.detach(), gout, ref, dims[7], dims)
    bf16 = dtype == torch.bfloat16
    _check('out', out, w_out, bf16)
    _check('grad_value', gv, w_gv, bf16)
    _check('d_offsets', d_off, w_off, bf16, smooth)
    _check('d_logits', d_logit, w_logit, bf16)


def test_tile_pass_overflowing_list_walks_all_queries():
    """Every sample of every query lands on ONE pixel of a 64x64 level: that tile's list overflows its
    capacity (4x the even load) and the workgroup walks all queries instead (csrc/msda_tile.hip, scan_all);
    the other tiles store zeros.  Plain fp32 Function and fused bf16 core against the oracle."""
    import MultiScaleDeformableAttention as MSDA
    from ops.functions import MSDeformAttnFusedFunction
    torch.manual_seed(3)
    N, M, D, P, L = 1, 2, 32, 4, 1
    shapes = [(64, 64)]
    S, Lq = 64 * 64, 5000
    hw = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
    lsi = cases.level_start_index(shapes).cuda()
    value = torch.randn(N, S, M, D, device='cuda')
    loc = (torch.tensor([0.31, 0.42], device='cuda') + 1e-3 * torch.rand(N, Lq, M, L, P, 2, device='cuda')).contiguous()
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, device='cuda'), -1).view(N, Lq, M, L, P).contiguous()
    gout = torch.randn(N, Lq, M * D, device='cuda')
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, hw, lsi, loc, attn, gout, 64)
    w_gv, w_gl, w_ga = oracle_msda.backward(value.double().cpu().numpy(), np.asarray(shapes, dtype=np.int64),
                                            oracle_msda.level_start_index(shapes), loc.double().cpu().numpy(),
                                            attn.double().cpu().numpy(), gout.double().cpu().numpy())
    scale = np.abs(w_gv).max()
    assert np.abs(gv.double().cpu().numpy() - w_gv).max() <= 1e-4 * scale     # ~5000 x 16 terms per pixel: relative to the sum
    assert (gv.abs().sum((0, 2, 3)) > 0).sum().item() <= 4                    # one 2x2 corner block, zeros elsewhere
    _check('grad_loc', gl, w_gl, False)
    _check('grad_attn', ga, w_ga, False)
    # fused bf16 core on the same pattern: reference points at the pixel, zero offsets
    ref = torch.tensor([0.31, 0.42], device='cuda').view(1, 1, 1, 2).repeat(1, Lq, 1, 1).contiguous()
    off = torch.zeros(N, Lq, M, L, P, 2, device='cuda', dtype=torch.bfloat16, requires_grad=True)
    logit = torch.randn(N, Lq, M, L * P, device='cuda').to(torch.bfloat16).requires_grad_(True)
    vb = value.to(torch.bfloat16).requires_grad_(True)
    out = MSDeformAttnFusedFunction.apply(vb, hw, lsi, off, logit, ref)
    gvb, = torch.autograd.grad(out, [vb], gout.to(torch.bfloat16))
    w = _oracle(vb.detach(), off.detach(), logit.detach(), gout.to(torch.bfloat16), ref, shapes,
                (N, M, D, P, Lq, L, S, shapes))
    _check('out (overflow case)', out, w[0], True)
    _check('grad_value (overflow case)', gvb, w[1], True)


@pytest.mark.parametrize('L,shapes,qshapes', [(1, [(16, 24)], [(32, 48), (16, 24), (8, 12)]), (3, [(32, 32), (16, 16), (8, 8)], [(16, 16)])])
def test_fused_core_is_capturable_with_fresh_geometry_tensors(L, shapes, qshapes):
    """forward (LDS-window schedule built on the device) and backward (tile pass planned on the
    device) read the level geometry from the DEVICE tensors and nothing else - no D2H copy, no cache keyed on tensor
    identity - so a HIP graph can capture them with geometry tensors that did not exist before the capture (the
    reference's deform_inputs makes new ones every forward, adapter_modules.py:28-47), and the replay equals eager."""
    from ops.functions import MSDeformAttnFusedFunction
    torch.manual_seed(3)
    N, M, D, P = 2, 6, 32, 4
    S, Lq = sum(h * w for h, w in shapes), sum(h * w for h, w in qshapes)
    value = torch.randn(N, S, M, D, device='cuda').bfloat16().requires_grad_(True)
    off = (cases.ring_offsets(M, L, P).cuda()[None, None] + torch.randn(N, Lq, M, L, P, 2, device='cuda')).bfloat16().requires_grad_(True)
    logit = torch.randn(N, Lq, M, L * P, device='cuda').bfloat16().requires_grad_(True)
    ref0 = cases.reference_grid(qshapes).cuda()
    hw0 = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
    lsi0 = cases.level_start_index(shapes).cuda()
    gout = torch.randn(N, Lq, M * D, device='cuda').bfloat16()

    def run():
        hw, lsi, ref = hw0.clone(), lsi0.clone(), ref0.clone()          # fresh tensors: new data_ptr, new identity
        out = MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref)
        return (out,) + torch.autograd.grad(out, [value, off, logit], gout)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            want = [t.float().clone() for t in run()]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = run()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for g, w, nm in zip(got, want, ('out', 'grad_value', 'd_offsets', 'd_logits')):
        # list order inside a tile depends on atomics: sums may differ in the last bits between two runs
        assert (g.float() - w).abs().max().item() <= 2e-2 * max(1.0, w.abs().max().item()), nm


def test_window_schedule_rides_on_the_reference_points_tensor():
    """ops/functions/ms_deform_attn_fused.py::_window_workspace: the extractor's six calls of one forward share the window
    schedule through an attribute on the reference_points tensor object.  Reused only for the same objects, unmodified, in
    the same pass; the result never depends on it (the schedule only decides which corner rows come from LDS)."""
    from ops.functions import ms_deform_attn_fused as mf
    torch.manual_seed(5)
    N, M, D, P, L = 2, 6, 32, 4, 1
    shapes, qshapes = [(16, 24)], [(32, 48), (16, 24), (8, 12)]
    S, Lq = 16 * 24, sum(h * w for h, w in qshapes)
    value = torch.randn(N, S, M, D, device='cuda').bfloat16()
    hw = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
    lsi = cases.level_start_index(shapes).cuda()
    ref = cases.reference_grid(qshapes).cuda()                      # (1, Lq, 1, 2): the tensor the module is called with

    def rows():
        y = torch.randn(N, Lq, M, 12, device='cuda') * 2
        return y[..., :8].unflatten(-1, (L, P, 2)), y[..., 8:]

    def call(offs, logs, carrier, token):
        return mf.fused_forward(value, hw, lsi, offs, logs, 12, 12, ref.float().contiguous().view(Lq, -1, 2), carrier=carrier, token=token)

    o1, l1 = rows()
    want1 = call(o1, l1, None, None)
    assert not hasattr(ref, '_vah_win_schedule')
    got1 = call(o1, l1, ref, 7)
    ws1 = ref._vah_win_schedule.ws
    assert torch.equal(got1, want1)
    o2, l2 = rows()
    got2 = call(o2, l2, ref, 7)                                      # same objects, same pass: only the forward kernel runs
    assert ref._vah_win_schedule.ws is ws1
    assert torch.equal(got2, call(o2, l2, None, None))
    got3 = call(o2, l2, ref, 9)                                      # another pass
    ws3 = ref._vah_win_schedule.ws
    assert ws3 is not ws1 and torch.equal(got3, got2)
    ref.copy_(ref.flip(1))                                           # written in place: the version counter moves
    want4 = call(o2, l2, None, None)
    got4 = call(o2, l2, ref, 9)
    assert ref._vah_win_schedule.ws is not ws3 and torch.equal(got4, want4)
    ws4 = ref._vah_win_schedule.ws
    hw2 = hw.clone()                                                 # a fresh geometry tensor
    got5 = mf.fused_forward(value, hw2, lsi, o2, l2, 12, 12, ref.float().contiguous().view(Lq, -1, 2), carrier=ref, token=9)
    assert ref._vah_win_schedule.ws is not ws4 and torch.equal(got5, want4)
    side = torch.cuda.Stream()                                       # another stream
    side.wait_stream(torch.cuda.current_stream())
    ws5 = ref._vah_win_schedule.ws
    with torch.cuda.stream(side):
        got6 = mf.fused_forward(value, hw2, lsi, o2, l2, 12, 12, ref.float().contiguous().view(Lq, -1, 2), carrier=ref, token=9)
    torch.cuda.current_stream().wait_stream(side)
    assert ref._vah_win_schedule.ws is not ws5 and torch.equal(got6, want4)
