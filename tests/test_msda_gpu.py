"""GPU parity of the HIP multi-scale deformable attention (through the C ABI, via the
MultiScaleDeformableAttention binding) against
  (a) the golden vectors made from the reference's Python (tests/golden/msda_*.npz), and
  (b) the C oracle on seeded inputs at the adapter's call shapes,
plus size-independent properties at the BASELINE full sizes.

Tolerances: fp32 1e-4 relative to max(1, |ref|_inf) (north-star: "within 1e-4 fp32"; the
reference's own fp32 check is rtol 1e-2 / atol 1e-3, detection/ops/test.py:68); fp64 1e-10.
grad_value is accumulated with float atomics (as in the reference), so bitwise equality is
never asserted.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cases, msda as oracle_msda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def MSDA():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    import MultiScaleDeformableAttention as m
    return m


@pytest.fixture(scope='module')
def Function():
    from ops.functions import MSDeformAttnFunction
    return MSDeformAttnFunction


def _dev(ts, dtype=None):
    out = []
    for t in ts:
        if t.dtype.is_floating_point and dtype is not None:
            t = t.to(dtype)
        out.append(t.cuda().contiguous())
    return out


def _run_hip(MSDA, value, hw, lsi, loc, attn, gout, dtype):
    v, s, i, l, a, g = _dev((value, hw, lsi, loc, attn, gout), dtype)
    out = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    torch.cuda.synchronize()
    return [t.double().cpu().numpy() for t in (out, gv, gl, ga)]


def _gate_mask(loc, hw):
    # see tests/test_oracle_golden.py::_gate_mask: samples exactly on the -1 gate
    norm = np.stack([hw.numpy()[:, 1], hw.numpy()[:, 0]], -1).astype(np.float64)
    px = loc.numpy().astype(np.float64) * norm[None, None, None, :, None, :] - 0.5
    return np.where((px == -1.0).any(-1, keepdims=True), 0.0, 1.0)


def _assert_close(got, ref, tol, what, mask=None):
    assert got.shape == ref.shape, what
    err = np.abs(got - ref)
    if mask is not None:
        err = err * mask
    bound = tol * max(1.0, np.abs(ref).max())
    assert np.isfinite(got).all(), what
    assert err.max() <= bound, '%s: max err %.3e > %.3e' % (what, err.max(), bound)


@pytest.mark.parametrize('name', sorted(cases.ADAPTER_CASES))
def test_hip_f32_matches_reference_goldens(MSDA, golden_dir, name):
    g2 = np.load(os.path.join(golden_dir, 'msda_adapter.npz'))
    value, hw, lsi, loc, attn, gout = cases.msda_inputs(name, **cases.ADAPTER_CASES[name])
    got = _run_hip(MSDA, value, hw, lsi, loc, attn, gout, torch.float32)
    for nm, x in zip(('out', 'gv', 'gl', 'ga'), got):
        _assert_close(x, g2['%s_%s' % (name, nm)].astype(np.float64), 1e-4, name + ':' + nm,
                      _gate_mask(loc, hw) if nm == 'gl' else None)


@pytest.mark.parametrize('D', cases.TESTPY_CHANNELS)
def test_hip_f64_matches_reference_testpy_goldens(MSDA, golden_dir, D):
    """The reference's only test (detection/ops/test.py): fp64 forward + every backward
    kernel variant's channel count."""
    g1 = np.load(os.path.join(golden_dir, 'msda_testpy.npz'))
    value, hw, lsi, loc, attn, gout = cases.testpy_inputs(D)
    got = _run_hip(MSDA, value, hw, lsi, loc, attn, gout, torch.float64)
    for nm, x in zip(('out', 'gv', 'gl', 'ga'), got):
        _assert_close(x, g1['D%d_%s' % (D, nm)], 1e-10, 'D%d:%s' % (D, nm))


@pytest.mark.parametrize('D', [30, 32, 64, 71])
def test_hip_f32_testpy_shapes(MSDA, golden_dir, D):
    """fp32 forward at the reference test's shapes (test.py:53-75 uses rtol 1e-2/atol 1e-3)."""
    g1 = np.load(os.path.join(golden_dir, 'msda_testpy.npz'))
    value, hw, lsi, loc, attn, gout = cases.testpy_inputs(D)
    got = _run_hip(MSDA, value, hw, lsi, loc, attn, gout, torch.float32)
    for nm, x in zip(('out', 'gv', 'gl', 'ga'), got):
        _assert_close(x, g1['D%d_%s' % (D, nm)], 1e-4, 'D%d:%s' % (D, nm))


@pytest.mark.parametrize('name', sorted(cases.PARITY_CASES))
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_hip_matches_oracle(MSDA, name, dtype):
    value, hw, lsi, loc, attn, gout = cases.msda_inputs(name, **cases.PARITY_CASES[name])
    got = _run_hip(MSDA, value, hw, lsi, loc, attn, gout, dtype)
    args = [value.double().numpy(), hw.numpy(), lsi.numpy(), loc.double().numpy(),
            attn.double().numpy()]
    ref = [oracle_msda.forward(*args)] + list(oracle_msda.backward(*args, gout.double().numpy()))
    tol = 1e-4 if dtype == torch.float32 else 1e-10
    for nm, x, r in zip(('out', 'gv', 'gl', 'ga'), got, ref):
        _assert_close(x, r, tol, name + ':' + nm)


def test_gradcheck_fp64(Function):
    """torch.autograd.gradcheck through the autograd Function, as detection/ops/test.py:78-101."""
    for D in (30, 32, 64, 71):
        value, hw, lsi, loc, attn, _ = cases.testpy_inputs(D)
        v, s, i, l, a = _dev((value, hw, lsi, loc, attn))
        v.requires_grad_(True), l.requires_grad_(True), a.requires_grad_(True)
        assert torch.autograd.gradcheck(Function.apply, (v, s, i, l, a, 2), nondet_tol=1e-9)


def test_autograd_function_contract(Function):
    """Backward returns grads for value / loc / attn only; AMP inputs are cast to fp32
    (ms_deform_attn_func.py:21,46)."""
    value, hw, lsi, loc, attn, gout = cases.msda_inputs('inj128_adapter',
                                                        **cases.PARITY_CASES['inj128_adapter'])
    v, s, i, l, a, g = _dev((value, hw, lsi, loc, attn, gout))
    v.requires_grad_(True), l.requires_grad_(True), a.requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.float16):
        out = Function.apply(v.half(), s, i, l.half(), a.half(), 64)
    assert out.dtype == torch.float32
    out = Function.apply(v, s, i, l, a, 64)
    out.backward(g)
    assert v.grad.shape == v.shape and l.grad.shape == l.shape and a.grad.shape == a.shape
    assert s.grad is None and i.grad is None


def test_error_behaviour(MSDA):
    value, hw, lsi, loc, attn, gout = cases.msda_inputs('inj128_uniform',
                                                        **cases.PARITY_CASES['inj128_uniform'])
    v, s, i, l, a, g = _dev((value, hw, lsi, loc, attn, gout))
    with pytest.raises(RuntimeError, match='contiguous'):
        MSDA.ms_deform_attn_forward(v.transpose(1, 2), s, i, l, a, 64)
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        MSDA.ms_deform_attn_forward(v, s.cpu(), i, l, a, 64)
    with pytest.raises(RuntimeError, match='CPU'):
        MSDA.ms_deform_attn_forward(v.cpu(), s, i, l, a, 64)
    v3 = torch.cat([v, v[:1]], 0)
    l3, a3 = torch.cat([l, l[:1]], 0), torch.cat([a, a[:1]], 0)
    with pytest.raises(RuntimeError, match='im2col_step'):
        MSDA.ms_deform_attn_forward(v3, s, i, l3, a3, 2)          # 3 % 2 != 0
    MSDA.ms_deform_attn_forward(v3, s, i, l3, a3, 64)            # min(3, 64) = 3 divides
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.half(), s, i, l.half(), a.half(), 64)


def test_empty_and_malformed_levels(MSDA):
    dev = 'cuda'
    s = torch.tensor([[2, 2]], dtype=torch.long, device=dev)
    i = torch.tensor([0], dtype=torch.long, device=dev)
    v = torch.randn(1, 4, 2, 32, device=dev)
    out = MSDA.ms_deform_attn_forward(v, s, i, torch.zeros(1, 0, 2, 1, 4, 2, device=dev),
                                      torch.zeros(1, 0, 2, 1, 4, device=dev), 64)
    assert out.shape == (1, 0, 64)
    # a level that claims more rows than the value tensor holds contributes nothing (no fault)
    bad = torch.tensor([[1000, 1000]], dtype=torch.long, device=dev)
    loc = torch.rand(1, 5, 2, 1, 4, 2, device=dev)
    att = torch.rand(1, 5, 2, 1, 4, device=dev)
    out = MSDA.ms_deform_attn_forward(v, bad, i, loc, att, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, bad, i, loc, att, torch.ones_like(out), 64)
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0 and float(gv.abs().max()) == 0
    assert float(gl.abs().max()) == 0 and float(ga.abs().max()) == 0


@pytest.mark.parametrize('geometry', ['trailing_rows', 'gap_between_levels', 'overlapping_levels'])
def test_levels_that_do_not_tile_the_value_rows(MSDA, geometry):
    """the tile pass STORES grad_value, so value rows no level covers must come back as zeros (the reference
    starts from zeros, ms_deform_attn_cuda.cu:121) and rows two levels share must receive both contributions (the
    reference adds atomically, cuh:87-159).  The plan kernel detects such geometry on the device, zero-fills and the
    tile pass adds instead of storing.  Against the C oracle (which follows level_start_index literally); D = 32,
    P = 4: the tiled path.  Also the fused bf16 core (grad_value only, against the plain fp32 path)."""
    torch.manual_seed(7)
    N, M, D, P, Lq = 2, 3, 32, 4, 50
    shapes = [(6, 4), (3, 2)]
    if geometry == 'trailing_rows':
        lsi, S = [0, 24], 40                    # rows 30..39 belong to no level
    elif geometry == 'gap_between_levels':
        lsi, S = [0, 30], 36                    # rows 24..29 belong to no level
    else:
        lsi, S = [0, 20], 26                    # rows 20..23 belong to both levels
    L = len(shapes)
    hw = torch.tensor(shapes, dtype=torch.long)
    li = torch.tensor(lsi, dtype=torch.long)
    value = torch.randn(N, S, M, D)
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    gout = torch.randn(N, Lq, M * D)
    v, s_, i_, l_, a_, g_ = _dev((value, hw, li, loc, attn, gout))
    out = MSDA.ms_deform_attn_forward(v, s_, i_, l_, a_, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s_, i_, l_, a_, g_, 64)
    torch.cuda.synchronize()
    args = [value.numpy(), hw.numpy(), li.numpy(), loc.numpy(), attn.numpy()]
    want = [oracle_msda.forward(*args)] + list(oracle_msda.backward(*args, gout.numpy()))
    for nm, x, r in zip(('out', 'gv', 'gl', 'ga'), (out, gv, gl, ga), want):
        _assert_close(x.cpu().numpy(), r, 1e-4, geometry + ':' + nm)
    if geometry != 'overlapping_levels':
        covered = torch.zeros(S, dtype=torch.bool)
        for (h, w), st in zip(shapes, lsi):
            covered[st:st + h * w] = True
        assert float(gv.cpu()[:, ~covered].abs().max()) == 0.0


def test_non_default_stream(MSDA):
    value, hw, lsi, loc, attn, gout = cases.msda_inputs('ext128_adapter',
                                                        **cases.PARITY_CASES['ext128_adapter'])
    v, s, i, l, a, g = _dev((value, hw, lsi, loc, attn, gout))
    ref = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    st.synchronize()
    assert torch.equal(out, ref)      # forward has no atomics: bitwise repeatable


# ---------------------------------------------------------------------------------------
# full BASELINE sizes: properties instead of stored outputs
# ---------------------------------------------------------------------------------------
def _full_inputs(cfg, mode, seed=0):
    N, M, D, P, Lq, shapes, qshapes = cases.bench_inputs(cfg)
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    L, S = len(shapes), sum(h * w for h, w in shapes)
    hw = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
    lsi = cases.level_start_index(shapes).cuda()
    value = torch.randn(N, S, M, D, device='cuda', generator=g)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, device='cuda', generator=g), -1).view(N, Lq, M, L, P)
    if mode == 'uniform':
        loc = torch.rand(N, Lq, M, L, P, 2, device='cuda', generator=g) * 1.2 - 0.1
    else:
        ref = cases.reference_grid(qshapes).cuda()
        off = cases.ring_offsets(M, L, P).cuda()[None, None] + torch.randn(
            N, Lq, M, L, P, 2, device='cuda', generator=g)
        wh = hw.flip(-1).float()
        loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    gout = torch.randn(N, Lq, M * D, device='cuda', generator=g)
    return value, hw, lsi, loc.contiguous(), attn.contiguous(), gout


@pytest.mark.parametrize('cfg', ['cfg1', 'cfg2_inj', 'cfg2_ext', 'cfg3_inj', 'cfg3_ext',
                                 'cfg4_inj', 'cfg4_ext', 'cfg5_inj', 'cfg5_ext', 'cfg5_pixdec'])
@pytest.mark.parametrize('mode', ['uniform', 'adapter'])
def test_full_size_properties(MSDA, cfg, mode):
    v, s, i, l, a, g = _full_inputs(cfg, mode)
    out = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    # (1) the op is linear in value: adjoint identity <out(v), g> == <v, grad_value(g)>
    # (both sides are sums of millions of signed fp32-accurate terms: the bound is relative to the sum of their
    # magnitudes, 1e-6 of it - a few fp32 ulps per term -, not to the cancelled total)
    lhs = (out.double() * g.double()).sum().item()
    rhs = (v.double() * gv.double()).sum().item()
    mag = (out.double() * g.double()).abs().sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)) + 1e-6 * mag, (lhs, rhs, mag)
    # (2) ... and linear in attn: <out, g> == <attn, grad_attn>
    rhs2 = (a.double() * ga.double()).sum().item()
    assert abs(lhs - rhs2) <= 1e-5 * max(1.0, abs(lhs)) + 1e-6 * mag, (lhs, rhs2, mag)
    # (3) linearity of the forward in value
    v2 = torch.randn_like(v)
    o2 = MSDA.ms_deform_attn_forward(v2, s, i, l, a, 64)
    o12 = MSDA.ms_deform_attn_forward(0.5 * v - 2.0 * v2, s, i, l, a, 64)
    assert (o12 - (0.5 * out - 2.0 * o2)).abs().max().item() <= 2e-4
    # (4) constant maps: every fully-inside sample returns the constant, so
    #     out == c * sum of the weights of samples whose 4 corners are inside
    c = torch.ones_like(v)
    oc = MSDA.ms_deform_attn_forward(c, s, i, l, a, 64)
    wh = s.flip(-1).float()
    px = l * wh[None, None, None, :, None, :] - 0.5
    full = ((px >= 0) & (px <= wh[None, None, None, :, None, :] - 1)).all(-1)
    part = ((px > -1) & (px < wh[None, None, None, :, None, :])).all(-1) & ~full
    lo = (a * full).sum((-1, -2))
    hi = lo + (a * part).sum((-1, -2))
    oc = oc.view(*lo.shape, -1)
    assert (oc >= lo[..., None] - 1e-4).all() and (oc <= hi[..., None] + 1e-4).all()
    # (5) forward is bitwise repeatable; backward loc/attn grads are too (no atomics there)
    out_b = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv_b, gl_b, ga_b = MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    assert torch.equal(out, out_b) and torch.equal(gl, gl_b) and torch.equal(ga, ga_b)
    assert (gv - gv_b).abs().max().item() <= 1e-3


def test_pixel_decoder_full_size_vs_oracle(MSDA):
    """BASELINE configs[4] call shape (Mask2Former pixel decoder: Lq = S = 22050, 8 heads, 3 levels, batch 2 -
    /root/reference/segmentation/mmseg_custom/models/plugins/msdeformattn_pixel_decoder.py:224-242) through the
    plain fp32 ms_deform_attn_forward / _backward (the tile-pass backward: no reference grid needed) against the C
    oracle at the SAME size in fp64: every element of out, grad_value, grad_loc, grad_attn."""
    v, s, i, l, a, g = _full_inputs('cfg5_pixdec', 'adapter')
    out = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    args = [v.double().cpu().numpy(), s.cpu().numpy(), i.cpu().numpy(), l.double().cpu().numpy(), a.double().cpu().numpy()]
    ref_out = oracle_msda.forward(*args)
    ref_gv, ref_gl, ref_ga = oracle_msda.backward(*args, g.double().cpu().numpy())
    mask = _gate_mask_np(l.double().cpu().numpy(), s.cpu().numpy())
    for nm, got, want in (('out', out, ref_out), ('grad_value', gv, ref_gv), ('grad_attn', ga, ref_ga)):
        err = np.abs(got.double().cpu().numpy() - want).max()
        assert err <= 1e-4 * max(1.0, np.abs(want).max()), (nm, err)
    err = np.abs(np.where(mask, gl.double().cpu().numpy() - ref_gl, 0.0)).max()
    assert err <= 1e-4 * max(1.0, np.abs(ref_gl).max()), ('grad_loc', err)


def _gate_mask_np(loc, shapes):
    """False where a pixel coordinate sits within 1e-4 of an integer: d(out)/d(loc) jumps there (bilinear kink) and
    the fp32 kernel / fp64 oracle may fall on different sides."""
    wh = shapes[:, ::-1].astype(np.float64)
    px = loc * wh[None, None, None, :, None, :] - 0.5
    ok = (np.abs(px - np.round(px)) > 1e-4).all(-1, keepdims=True)
    return np.broadcast_to(ok, loc.shape)


def test_full_size_spot_check_vs_oracle(MSDA):
    """cfg3 injector at full size: a random subset of queries is recomputed by the C oracle."""
    v, s, i, l, a, g = _full_inputs('cfg3_inj', 'adapter')
    out = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
    idx = torch.randperm(l.shape[1], generator=torch.Generator().manual_seed(1))[:97]
    ls, as_, gs = l[:, idx.cuda()], a[:, idx.cuda()], g[:, idx.cuda()]
    args = [v.double().cpu().numpy(), s.cpu().numpy(), i.cpu().numpy(),
            ls.double().cpu().numpy(), as_.double().cpu().numpy()]
    ref_out = oracle_msda.forward(*args)
    _, ref_gl, ref_ga = oracle_msda.backward(*args, gs.double().cpu().numpy())
    _assert_close(out[:, idx.cuda()].double().cpu().numpy(), ref_out, 1e-4, 'out subset')
    _assert_close(gl[:, idx.cuda()].double().cpu().numpy(), ref_gl, 1e-4, 'gl subset')
    _assert_close(ga[:, idx.cuda()].double().cpu().numpy(), ref_ga, 1e-4, 'ga subset')
