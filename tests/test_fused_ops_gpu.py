"""GPU: the fused memory-bound operators (csrc/fused_ops.hip) against the PyTorch expressions they
replace (the reference's module arithmetic under bf16 autocast).  Tolerance: bf16 rounding of the
outputs (2^-8 relative) for tensors, fp32 accumulation noise for parameter gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    err = (a.float() - b.float()).abs().max().item()
    ref = max(1.0, b.float().abs().max().item())
    assert err <= tol * ref, '%s: %.3e (ref %.3e)' % (what, err, ref)


@pytest.mark.parametrize('shape', [(2, 100, 192), (3, 37, 768), (1, 513, 1024), (2, 7, 64), (1, 5, 2048)])
def test_layer_norm_bf16(shape):
    from vitadapter import fused
    torch.manual_seed(0)
    ln = torch.nn.LayerNorm(shape[-1], eps=1e-6).cuda()
    with torch.no_grad():
        ln.weight.normal_(1, 0.2)
        ln.bias.normal_(0, 0.2)
    x = (torch.randn(shape, device='cuda') * 2 + 0.5).requires_grad_(True)
    g = torch.randn(shape, device='cuda').to(torch.bfloat16)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = fused.layer_norm(ln, x)
    assert y.dtype == torch.bfloat16
    y.backward(g)
    gx, gw, gb = x.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()
    x.grad = None
    ln.zero_grad()
    yr = ln(x)
    yr.backward(g.float())
    _close(y, yr, 1e-2, 'y')
    _close(gx, x.grad, 1e-4, 'dx')
    _close(gw, ln.weight.grad, 1e-3, 'dw')
    _close(gb, ln.bias.grad, 1e-3, 'db')
    assert fused.layer_norm(ln, x).dtype == torch.float32          # no autocast -> plain LayerNorm


@pytest.mark.parametrize('with_gamma,prob', [(True, 0.3), (True, 0.0), (False, 0.4), (False, 0.0)])
def test_scale_residual(with_gamma, prob):
    from vitadapter import fused
    from vitadapter.backbones.vit import DropPath
    torch.manual_seed(1)
    B, N, C = 4, 300, 768
    x = torch.randn(B, N, C, device='cuda', requires_grad=True)
    z = torch.randn(B, N, C, device='cuda').to(torch.bfloat16).requires_grad_(True)
    gamma = (torch.randn(C, device='cuda') * 0.5).requires_grad_(True) if with_gamma else None
    dp = DropPath(prob).train()
    torch.manual_seed(5)
    y = fused.residual(x, z, gamma, dp)
    g = torch.randn_like(y)
    y.backward(g)
    got = [x.grad.clone(), z.grad.clone()] + ([gamma.grad.clone()] if with_gamma else [])
    # the same mask: replay the RNG the helper consumed
    torch.manual_seed(5)
    keep = 1 - prob
    s = x.new_empty((B,)).bernoulli_(keep).div_(keep) if prob > 0 else torch.ones(B, device='cuda')
    x2, z2 = x.detach().clone().requires_grad_(True), z.detach().clone().requires_grad_(True)
    g2 = gamma.detach().clone().requires_grad_(True) if with_gamma else None
    t = (g2 * z2) if with_gamma else z2.float()
    yr = x2 + t * s.view(B, 1, 1)
    yr.backward(g)
    _close(y, yr, 1e-6, 'y')
    _close(got[0], x2.grad, 1e-6, 'dx')
    _close(got[1], z2.grad, 1e-2, 'dz (bf16)')
    if with_gamma:
        _close(got[2], g2.grad, 1e-4, 'dgamma')


@pytest.mark.parametrize('B,H,W,C', [(2, 4, 6, 48), (1, 8, 8, 192), (2, 2, 2, 16)])
def test_dwconv_tokens(B, H, W, C):
    from vitadapter import fused
    from vitadapter.backbones.adapter_modules import DWConv
    torch.manual_seed(2)
    m = DWConv(C).cuda()
    n = (H // 2) * (W // 2)
    x = torch.randn(B, 21 * n, C, device='cuda').to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(B, 21 * n, C, device='cuda').to(torch.bfloat16)
    y = m(x, H, W)                                # fused path (bf16 cuda input)
    y.backward(g)
    got = [y, x.grad.clone(), m.dwconv.weight.grad.clone(), m.dwconv.bias.grad.clone()]
    x.grad = None
    m.zero_grad()
    fused.ENABLED['dwconv'] = False
    try:
        xf = x.detach().float().requires_grad_(True)
        yr = m(xf, H, W)                          # reference arithmetic: slice / conv2d / cat in fp32
        yr.backward(g.float())
    finally:
        fused.ENABLED['dwconv'] = True
    _close(got[0], yr, 1e-2, 'y')
    _close(got[1], xf.grad, 1e-2, 'dx')
    _close(got[2], m.dwconv.weight.grad, 2e-3, 'dw')
    _close(got[3], m.dwconv.bias.grad, 2e-3, 'db')


def test_block_under_autocast_matches_unfused():
    """A whole ViT block + Extractor under bf16 autocast: fused vs unfused graph."""
    from vitadapter import fused
    from vitadapter.backbones import vit
    torch.manual_seed(3)
    blk = vit.Block(dim=192, num_heads=3, mlp_ratio=4., qkv_bias=True, layer_scale=True,
                    drop_path=0.0, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).cuda()
    with torch.no_grad():
        blk.gamma1.normal_(1, 0.3)
        blk.gamma2.normal_(1, 0.3)
    x = torch.randn(2, 8 * 8, 192, device='cuda')
    outs = []
    for on in (True, False):
        for k in fused.ENABLED:
            fused.ENABLED[k] = on
        xx = x.clone().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = blk(xx, 8, 8)
        y.float().square().mean().backward()
        outs.append((y.detach(), xx.grad.detach(), blk.gamma1.grad.clone(), blk.norm1.weight.grad.clone()))
        blk.zero_grad()
    for k in fused.ENABLED:
        fused.ENABLED[k] = True
    for a, b, nm in zip(outs[0], outs[1], ('y', 'dx', 'dgamma1', 'dnorm1.w')):
        _close(a, b, 3e-2, nm)


@pytest.mark.parametrize('shape', [(2, 100, 192), (1, 513, 768)])
def test_layer_norm_keep_sums_both_gradients(shape):
    """x + branch(LN(x)): the residual gradient and the LayerNorm gradient are summed inside the
    backward kernel; result = what autograd's separate add gives."""
    from vitadapter import fused
    torch.manual_seed(3)
    ln = torch.nn.LayerNorm(shape[-1], eps=1e-6).cuda()
    x = (torch.randn(shape, device='cuda') * 2 + 0.5).requires_grad_(True)
    g_res = torch.randn(shape, device='cuda')
    g_ln = torch.randn(shape, device='cuda').to(torch.bfloat16)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        xk, h = fused.layer_norm_keep(ln, x * 1.0)
    assert h.dtype == torch.bfloat16 and xk.dtype == torch.float32
    torch.autograd.backward([xk, h], [g_res, g_ln])
    got = x.grad.clone()
    gw = ln.weight.grad.clone()
    x.grad = None
    ln.zero_grad()
    xr = x * 1.0
    torch.autograd.backward([xr, ln(xr)], [g_res, g_ln.float()])
    _close(got, x.grad, 1e-4, 'dx')
    _close(gw, ln.weight.grad, 1e-3, 'dw')
    # only one of the two outputs used
    x.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        xk, h = fused.layer_norm_keep(ln, x * 1.0)
    xk.backward(g_res)
    _close(x.grad, g_res, 1e-7, 'dx (residual only)')
    x.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        xk, h = fused.layer_norm_keep(ln, x * 1.0)
    h.backward(g_ln)
    got = x.grad.clone()
    x.grad = None
    ln(x).backward(g_ln.float())
    _close(got, x.grad, 1e-4, 'dx (LN only)')


@pytest.mark.parametrize('rows,C', [(8192, 768), (43008, 96), (100, 2304), (1, 8), (33, 3072), (0, 64)])
def test_colsum_bf16(rows, C):
    import _vah
    torch.manual_seed(4)
    g = torch.randn(rows, C, device='cuda').to(torch.bfloat16)
    out = torch.full((C,), float('nan'), device='cuda')
    ws = torch.empty(_vah.lib.vah_reduce_ws_floats(C), device='cuda')
    _vah.check(_vah.lib.vah_colsum_bf16(g.data_ptr(), rows, C, out.data_ptr(), ws.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), 'colsum')
    ref = g.double().sum(0)
    assert (out.double() - ref).abs().max().item() <= 1e-5 * max(1.0, rows ** 0.5) * 4


@pytest.mark.parametrize('shape,out_f,bias', [((2, 300, 768), 2304, True), ((2, 300, 768), 768, False),
                                             ((1, 21, 192), 96, True), ((64, 3072), 768, True)])
def test_linear_bf16_matches_autocast(shape, out_f, bias):
    """fused.linear = F.linear under bf16 autocast: same forward bits, input gradient to bf16
    rounding, weight / bias gradients at least as close to the fp32 result as autocast's."""
    from vitadapter import fused
    torch.manual_seed(5)
    lin = torch.nn.Linear(shape[-1], out_f, bias=bias).cuda()
    x = torch.randn(shape, device='cuda').to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(shape[:-1] + (out_f,), device='cuda').to(torch.bfloat16)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = fused.linear(lin, x)
    assert y.dtype == torch.bfloat16
    y.backward(g)
    got = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone()] + ([lin.bias.grad.clone()] if bias else [])
    assert lin.weight.grad.dtype == torch.float32
    x.grad = None
    lin.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        yr = lin(x)
    yr.backward(g)
    _close(got[0], yr, 1e-2, 'y')
    _close(got[1], x.grad, 1e-2, 'dx')
    # fp32 truth of the parameter gradients from the same bf16 operands
    g2, x2 = g.reshape(-1, out_f).double(), x.detach().reshape(-1, shape[-1]).double()
    gw = g2.t() @ x2
    e_ours = (got[2].double() - gw).abs().max().item()
    e_amp = (lin.weight.grad.double() - gw).abs().max().item()
    assert e_ours <= max(e_amp, 1e-3 * gw.abs().max().item()), (e_ours, e_amp)
    if bias:
        gb = g2.sum(0)
        assert (got[3].double() - gb).abs().max().item() <= 1e-4 * max(1.0, gb.abs().max().item())
    # parameters updated in place (optimizer step): the bf16 copy follows
    with torch.no_grad():
        lin.weight.mul_(0.5)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        with fused.forward_epoch(lin):
            y2 = fused.linear(lin, x)
        y2r = lin(x)
    _close(y2, y2r, 1e-2, 'y after update')
    assert fused.linear(lin, x.float()).dtype == torch.float32       # no autocast -> plain nn.Linear


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K,f32', [(300, 768, 192, False), (768, 2304, 600, True), (8, 8, 8, False),
                                       (129, 96, 1000, True)])
def test_gemm_bf16_dispatcher(ta, tb, M, N, K, f32):
    """csrc/gemm.hip against torch.matmul in fp64 on the same bf16 operands."""
    from vitadapter import fused
    torch.manual_seed(6)
    a = torch.randn((K, M) if ta else (M, K), device='cuda').to(torch.bfloat16)
    b = torch.randn((N, K) if tb else (K, N), device='cuda').to(torch.bfloat16)
    bias = torch.randn(N, device='cuda')
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    d = fused.gemm_bf16(a, b, ta, tb, torch.float32 if f32 else torch.bfloat16)
    tol = (1e-5 if f32 else 8e-3) * max(1.0, ref.abs().max().item())
    assert (d.double() - ref).abs().max().item() <= tol
    d = fused.gemm_bf16(a, b, ta, tb, torch.float32 if f32 else torch.bfloat16, bias=bias)
    assert (d.double() - (ref + bias.double())).abs().max().item() <= tol
    d2 = fused.gemm_bf16(a, b, ta, tb, torch.float32 if f32 else torch.bfloat16, bias=bias)
    assert torch.equal(d, d2)                      # cached algorithm: deterministic replay


@pytest.mark.parametrize('Co,Ci,HW', [(1024, 64, 200 * 336), (1024, 64, 128 * 128), (768, 64, 128 * 128)])
def test_gemm_every_element_written_on_a_dirty_workspace(Co, Ci, HW):
    """spm.fc1 of ViT-Adapter-L as plane GEMMs (Co x HW x Ci): round 3 found library algorithms for it that leave most
    of the output untouched once the per-call workspace holds another tensor's bytes, while their first columns are
    right.  The output is pre-filled with NaN and the allocator's blocks are dirtied between calls."""
    from vitadapter import fused
    torch.manual_seed(11)
    w = torch.randn(Co, Ci, device='cuda').to(torch.bfloat16)
    x = torch.randn(HW, Ci, device='cuda').to(torch.bfloat16)
    ref = w.float() @ x.float().t()
    for it in range(4):
        junk = torch.full((48 << 20,), 0x7F + it, dtype=torch.uint8, device='cuda')
        del junk
        out = torch.full((Co, HW), float('nan'), dtype=torch.bfloat16, device='cuda')
        fused.gemm_bf16(w, x, trans_b=True, out=out)
        assert bool(torch.isfinite(out).all()), 'call %d left %d elements unwritten' % (it, int((~torch.isfinite(out)).sum()))
        assert (out.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()


def test_gemm_table_roundtrip():
    import _vah
    from vitadapter import fused
    a = torch.randn(64, 128, device='cuda').to(torch.bfloat16)
    b = torch.randn(128, 256, device='cuda').to(torch.bfloat16)
    d = fused.gemm_bf16(a, b)
    text = _vah.gemm_table_dump()
    line = [ln for ln in text.splitlines() if ln.startswith('0 0 0 0 0 64 256 128 ')]
    assert len(line) == 1, text
    assert text.startswith('#hipblaslt ')
    assert _vah.gemm_table_load(text) == len(text.splitlines()) - 1
    assert _vah.gemm_table_load('#hipblaslt 1\n' + text.split('\n', 1)[1]) == 0      # other build: ignored
    assert torch.equal(fused.gemm_bf16(a, b), d)   # entries are re-resolved from their index
    with pytest.raises(RuntimeError):
        _vah.gemm_table_load('not a table line')


@pytest.mark.parametrize('N,C,H,W,scale,with_b,a_bf16,training', [
    (2, 24, 64, 64, 4, True, True, True),      # norm1: up(c2) + c1 + interp(x, 4)
    (2, 16, 32, 48, 2, False, False, True),    # norm2: c2 + interp(x, 2)
    (1, 8, 16, 16, 1, False, False, True),     # norm3: c3 + x
    (2, 8, 96, 32, 4, True, True, False),      # eval: running statistics
    (1, 4, 128, 256, 8, False, True, True),    # several row tiles per plane
    (3, 5, 40, 24, 2, True, False, True),
])
def test_bn_tail_matches_reference_expression(N, C, H, W, scale, with_b, a_bf16, training):
    """fused.bn_tail = norm(a + b + F.interpolate(x, scale)) (vit_adapter.py:106-127) in fp32 on the
    same (bf16-valued) operands: outputs, all five gradients and the running statistics."""
    from vitadapter import fused
    torch.manual_seed(7)
    dt = torch.bfloat16 if a_bf16 else torch.float32
    a = torch.randn(N, C, H, W, device='cuda').to(dt).requires_grad_(True)
    b = (torch.randn(N, C, H, W, device='cuda') + 0.5).to(torch.bfloat16).requires_grad_(True) if with_b else None
    x = torch.randn(N, C, H // scale, W // scale, device='cuda', requires_grad=True)
    bn = torch.nn.BatchNorm2d(C).cuda()
    ref_bn = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.normal_(1, 0.3)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2)
    ref_bn.load_state_dict(bn.state_dict())
    bn.train(training)
    ref_bn.train(training)
    g = torch.randn(N, C, H, W, device='cuda')
    # a per-channel shift (folded conv biases) rides along in every case
    shift = (torch.randn(C, device='cuda') * 0.7).requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = fused.bn_tail(bn, a, b, x, scale, shift)
    assert y.dtype == torch.float32
    y.backward(g)
    got = dict(a=a.grad.clone(), x=x.grad.clone(), w=bn.weight.grad.clone(), bias=bn.bias.grad.clone(),
               shift=shift.grad.clone())
    if with_b:
        got['b'] = b.grad.clone()
    a2 = a.detach().float().requires_grad_(True)
    b2 = b.detach().float().requires_grad_(True) if with_b else None
    x2 = x.detach().clone().requires_grad_(True)
    shift2 = shift.detach().clone().requires_grad_(True)
    t = a2 + b2 if with_b else a2
    t = t + shift2.view(1, -1, 1, 1)
    t = t + (x2 if scale == 1 else F.interpolate(x2, scale_factor=scale, mode='bilinear', align_corners=False))
    yr = ref_bn(t)
    yr.backward(g)
    _close(y, yr, 2e-5, 'y')
    assert (got['shift'] - shift2.grad).abs().max().item() <= 2e-3 * max(1.0, g.abs().sum((0, 2, 3)).max().item() * 1e-2)
    _close(got['a'], a2.grad, 1e-2 if a_bf16 else 2e-5, 'da')
    if with_b:
        _close(got['b'], b2.grad, 1e-2, 'db')
    _close(got['x'], x2.grad, 5e-5, 'dx')
    _close(got['w'], ref_bn.weight.grad, 1e-4, 'dweight')
    _close(got['bias'], ref_bn.bias.grad, 1e-4, 'dbias')
    _close(bn.running_mean, ref_bn.running_mean, 1e-5, 'running_mean')
    _close(bn.running_var, ref_bn.running_var, 1e-5, 'running_var')
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)


@pytest.mark.parametrize('with_gamma,prob,C', [(True, 0.3, 768), (False, 0.0, 192), (True, 0.0, 1024), (False, 0.5, 64)])
def test_residual_ln_pair(with_gamma, prob, C):
    """fused.residual_ln = (t, norm(t)) with t = x + drop_path(gamma * z): forward and all gradients
    against the two separate ops in fp32, with both outputs feeding the loss."""
    from vitadapter import fused
    from vitadapter.backbones.vit import DropPath
    torch.manual_seed(8)
    B, N = 3, 157
    ln = torch.nn.LayerNorm(C, eps=1e-6).cuda()
    with torch.no_grad():
        ln.weight.normal_(1, 0.2)
        ln.bias.normal_(0, 0.2)
    x = torch.randn(B, N, C, device='cuda', requires_grad=True)
    z = torch.randn(B, N, C, device='cuda').to(torch.bfloat16).requires_grad_(True)
    gamma = (torch.randn(C, device='cuda') * 0.5).requires_grad_(True) if with_gamma else None
    dp = DropPath(prob).train()
    gt = torch.randn(B, N, C, device='cuda')
    gh = torch.randn(B, N, C, device='cuda').to(torch.bfloat16)
    torch.manual_seed(9)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        t, h = fused.residual_ln(x, z, gamma, dp, ln)
    assert type(t.grad_fn).__name__ == '_ResidualLNBackward' and h.dtype == torch.bfloat16
    torch.autograd.backward([t, h], [gt, gh])
    got = [x.grad.clone(), z.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()] + \
        ([gamma.grad.clone()] if with_gamma else [])
    torch.manual_seed(9)
    keep = 1 - prob
    s = x.new_empty((B,)).bernoulli_(keep).div_(keep) if prob > 0 else torch.ones(B, device='cuda')
    x2, z2 = x.detach().clone().requires_grad_(True), z.detach().float().requires_grad_(True)
    g2 = gamma.detach().clone().requires_grad_(True) if with_gamma else None
    ln.zero_grad()
    tr = x2 + ((g2 * z2) if with_gamma else z2) * s.view(B, 1, 1)
    hr = ln(tr)
    torch.autograd.backward([tr, hr], [gt, gh.float()])
    _close(t, tr, 1e-6, 't')
    _close(h, hr, 1e-2, 'h')
    _close(got[0], x2.grad, 1e-4, 'dx')
    _close(got[1], z2.grad, 1e-2, 'dz (bf16)')
    _close(got[2], ln.weight.grad, 1e-3, 'dw')
    _close(got[3], ln.bias.grad, 1e-3, 'db')
    if with_gamma:
        _close(got[4], g2.grad, 1e-3, 'dgamma')
    # only the residual output used
    x.grad = z.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        t, h = fused.residual_ln(x, z, gamma, None, ln)
    t.backward(gt)
    _close(x.grad, gt, 1e-7, 'dx (t only)')


@pytest.mark.parametrize('N,C,H,W,dt,training', [(2, 16, 64, 64, torch.bfloat16, True), (1, 8, 32, 128, torch.float32, True),
                                                  (2, 8, 48, 32, torch.bfloat16, False)])
def test_bn_relu_matches_reference_expression(N, C, H, W, dt, training, monkeypatch):
    """fused.bn_relu = relu(norm(a)) (the conv -> SyncBN -> ReLU triples of the SPM): output in a's dtype,
    gradients of a / weight / bias, running statistics."""
    from vitadapter import fused
    monkeypatch.setattr(fused, 'BN_RELU_MIN_NUMEL', 0)
    torch.manual_seed(9)
    a = (torch.randn(N, C, H, W, device='cuda') * 1.5 + 0.3).to(dt).requires_grad_(True)
    bn, ref_bn = torch.nn.BatchNorm2d(C).cuda(), torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.normal_(1, 0.3)
        bn.bias.normal_(0, 0.5)
        bn.running_mean.normal_(0.3, 0.2)
        bn.running_var.uniform_(1.5, 3)
    ref_bn.load_state_dict(bn.state_dict())
    bn.train(training)
    ref_bn.train(training)
    g = torch.randn(N, C, H, W, device='cuda').to(dt)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = fused.bn_relu(bn, a)
    assert y.dtype == dt and type(y.grad_fn).__name__ == '_BNTailBackward'
    y.backward(g)
    a2 = a.detach().float().requires_grad_(True)
    yr = F.relu(ref_bn(a2))
    yr.backward(g.float())
    tol = 1e-2 if dt == torch.bfloat16 else 3e-5
    _close(y, yr, tol, 'y')
    # entries whose pre-activation sits within rounding of 0 may take the other side of the ReLU
    diff = (a.grad.float() - a2.grad).abs()
    lim = tol * max(1.0, a2.grad.abs().max().item())
    assert (diff > lim).float().mean().item() <= (2e-3 if dt == torch.bfloat16 else 1e-4), diff.max().item()
    _close(bn.weight.grad, ref_bn.weight.grad, 2e-2 if dt == torch.bfloat16 else 2e-4, 'dweight')
    _close(bn.bias.grad, ref_bn.bias.grad, 2e-2 if dt == torch.bfloat16 else 2e-4, 'dbias')
    _close(bn.running_mean, ref_bn.running_mean, 1e-5, 'running_mean')
    _close(bn.running_var, ref_bn.running_var, 1e-5, 'running_var')


@pytest.mark.parametrize('B,C,H,W', [(2, 64, 8, 12), (1, 96, 6, 6), (3, 40, 4, 10)])
def test_tokens_to_maps(B, C, H, W):
    """fused.tokens_to_maps = the slice / transpose / reshape / contiguous lines of the pyramid assembly
    (vit_adapter.py:113-119), forward bit-exact, backward into one gradient tensor (one map unused)."""
    from vitadapter import fused
    torch.manual_seed(10)
    hw = [(2 * H, 2 * W), (H, W), (H // 2, W // 2)]
    T = sum(h * w for h, w in hw)
    c = torch.randn(B, T, C, device='cuda', requires_grad=True)
    maps = fused.tokens_to_maps(c, hw)
    assert type(maps[0].grad_fn).__name__ == '_TokensToMapsBackward'
    c2 = c.detach().clone().requires_grad_(True)
    ref, t0 = [], 0
    for h, w in hw:
        ref.append(c2[:, t0:t0 + h * w].transpose(1, 2).reshape(B, C, h, w).contiguous())
        t0 += h * w
    for m, r in zip(maps, ref):
        assert torch.equal(m, r)
    gs = [torch.randn_like(m) for m in maps]
    torch.autograd.backward([maps[0], maps[2]], [gs[0], gs[2]])            # the middle map gets no gradient
    torch.autograd.backward([ref[0], ref[2]], [gs[0], gs[2]])
    assert torch.equal(c.grad, c2.grad)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_maps_to_tokens(dt):
    """fused.maps_to_tokens = cat([map.flatten(2).transpose(1, 2) + vec]) in fp32 (SPM output,
    vit_adapter.py:94-97): forward and the gradients of maps and vectors."""
    from vitadapter import fused
    torch.manual_seed(11)
    B, C = 2, 72
    hw = [(8, 12), (4, 6), (2, 3)]
    maps = [torch.randn(B, C, h, w, device='cuda').to(dt).requires_grad_(True) for h, w in hw]
    vecs = [torch.randn(C, device='cuda', requires_grad=True) for _ in hw]
    out = fused.maps_to_tokens(maps, vecs)
    assert out.dtype == torch.float32 and type(out.grad_fn).__name__ == '_MapsToTokensBackward'
    g = torch.randn_like(out)
    out.backward(g)
    maps2 = [m.detach().clone().requires_grad_(True) for m in maps]
    vecs2 = [v.detach().clone().requires_grad_(True) for v in vecs]
    ref = torch.cat([m.flatten(2).transpose(1, 2).float() + v for m, v in zip(maps2, vecs2)], dim=1)
    ref.backward(g)
    assert torch.equal(out, ref)
    for m, m2 in zip(maps, maps2):
        assert m.grad.dtype == dt and torch.equal(m.grad, m2.grad)
    for v, v2 in zip(vecs, vecs2):
        _close(v.grad, v2.grad, 1e-5, 'dvec')


def test_linear_pair_matches_two_linears():
    """fused.linear_pair(a, b, x) = (a(x), b(x)) under bf16 autocast: forward and every gradient against the
    two separate fused linears (same bf16 operands, fp32 accumulation)."""
    from vitadapter import fused
    torch.manual_seed(13)
    a, b = torch.nn.Linear(768, 96).cuda(), torch.nn.Linear(768, 48).cuda()
    x = torch.randn(2, 301, 768, device='cuda').to(torch.bfloat16).requires_grad_(True)
    ga = torch.randn(2, 301, 96, device='cuda').to(torch.bfloat16)
    gb = torch.randn(2, 301, 48, device='cuda').to(torch.bfloat16)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ya, yb = fused.linear_pair(a, b, x)
    assert type(ya.grad_fn).__name__ != 'AddmmBackward0' and ya.shape == (2, 301, 96) and yb.shape == (2, 301, 48)
    torch.autograd.backward([ya, yb], [ga, gb])
    got = [ya.detach().clone(), yb.detach().clone(), x.grad.clone(), a.weight.grad.clone(), a.bias.grad.clone(),
           b.weight.grad.clone(), b.bias.grad.clone()]
    x.grad = None
    a.zero_grad()
    b.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ra, rb = fused.linear(a, x), fused.linear(b, x)
    torch.autograd.backward([ra, rb], [ga, gb])
    want = [ra, rb, x.grad, a.weight.grad, a.bias.grad, b.weight.grad, b.bias.grad]
    for g, w, nm, tol in zip(got, want, ('ya', 'yb', 'dx', 'dWa', 'dba', 'dWb', 'dbb'),
                             (1e-2, 1e-2, 2e-2, 1e-4, 1e-4, 1e-4, 1e-4)):
        _close(g, w, tol, nm)
    # the concatenated bf16 copy follows an in-place parameter update
    with torch.no_grad():
        b.weight.mul_(0.5)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        _, yb2 = fused.linear_pair(a, b, x)
        rb2 = fused.linear(b, x)
    _close(yb2, rb2, 1e-2, 'yb after update')


@pytest.mark.parametrize('N,C,H,W', [(2, 8, 32, 48), (1, 3, 17, 9), (2, 4, 64, 64)])
def test_max_pool_3x3_s2_matches_torch(N, C, H, W):
    """fused.max_pool = nn.MaxPool2d(3, 2, 1) on bf16 incl. the tie rule (inputs are post-ReLU: exact
    zeros repeat, the first maximum in scan order takes the gradient) - bit-exact both ways."""
    from vitadapter import fused
    torch.manual_seed(14)
    pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    x = torch.relu(torch.randn(N, C, H, W, device='cuda')).to(torch.bfloat16).requires_grad_(True)
    y = fused.max_pool(pool, x)
    assert type(y.grad_fn).__name__ == '_MaxPool3s2Backward'
    g = torch.randn_like(y)
    y.backward(g)
    x2 = x.detach().clone().requires_grad_(True)
    yr = pool(x2)
    yr.backward(g)
    assert torch.equal(y, yr)
    assert torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 64, 768, 32, 32), (1, 128, 96, 8, 24), (2, 256, 64, 4, 6)])
def test_conv1x1_as_plane_gemms(B, Ci, Co, H, W):
    """fused.conv1x1 = F.conv2d(x, w) for a 1x1 convolution under bf16 autocast (SPM fc1..fc4)."""
    from vitadapter import fused
    torch.manual_seed(15)
    conv = torch.nn.Conv2d(Ci, Co, 1, bias=True).cuda()
    x = torch.randn(B, Ci, H, W, device='cuda').to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(B, Co, H, W, device='cuda').to(torch.bfloat16)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = fused.conv1x1(conv, x)
    assert type(y.grad_fn).__name__ == '_Conv1x1BF16Backward' and y.dtype == torch.bfloat16
    y.backward(g)
    # fp64 statement of the same products on the same bf16 operands (MIOpen's fp32 convolution is not a
    # tight enough reference for the fp32 weight gradient)
    xd, gd = x.detach().double().flatten(2), g.double().flatten(2)
    wd = conv.weight.detach().to(torch.bfloat16).double().view(Co, Ci)
    yr = torch.einsum('oi,bip->bop', wd, xd).view(B, Co, H, W)
    dxr = torch.einsum('oi,bop->bip', wd, gd).view(B, Ci, H, W)
    dwr = torch.einsum('bop,bip->oi', gd, xd).view(Co, Ci, 1, 1)
    _close(y, yr, 1e-2, 'y')
    _close(x.grad, dxr, 1e-2, 'dx')
    _close(conv.weight.grad, dwr, 1e-5, 'dw')
    assert conv.weight.grad.dtype == torch.float32 and conv.weight.grad.shape == conv.weight.shape


@pytest.mark.parametrize('shape', [(2, 150, 768), (1, 77, 192), (3, 33, 1024)])
def test_layer_norm_dual_keep(shape):
    """fused.layer_norm_dual_keep = (x, norm_a(x), norm_b(x)) with shared statistics; backward = residual
    gradient + both LayerNorm gradients in one pass; also with one of the three outputs unused."""
    from vitadapter import fused
    torch.manual_seed(16)
    C = shape[-1]
    na, nb = torch.nn.LayerNorm(C, eps=1e-6).cuda(), torch.nn.LayerNorm(C, eps=1e-6).cuda()
    with torch.no_grad():
        for n in (na, nb):
            n.weight.normal_(1, 0.3)
            n.bias.normal_(0, 0.3)
    x = (torch.randn(shape, device='cuda') * 1.7 + 0.4).requires_grad_(True)
    gr = torch.randn(shape, device='cuda')
    ga = torch.randn(shape, device='cuda').to(torch.bfloat16)
    gb = torch.randn(shape, device='cuda').to(torch.bfloat16)
    for use in ((True, True, True), (False, True, True), (True, True, False)):
        x.grad = None
        na.zero_grad()
        nb.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            xk, ya, yb = fused.layer_norm_dual_keep(na, nb, x * 1.0)
        assert type(ya.grad_fn).__name__ == '_LayerNormDualBF16Backward'
        outs = [t for t, u in zip((xk, ya, yb), use) if u]
        grads = [g for g, u in zip((gr, ga, gb), use) if u]
        torch.autograd.backward(outs, grads)
        got = [x.grad.clone(), na.weight.grad.clone(), na.bias.grad.clone()] + \
            ([nb.weight.grad.clone(), nb.bias.grad.clone()] if use[2] else [])
        x.grad = None
        na.zero_grad()
        nb.zero_grad()
        xr = x * 1.0
        ref_outs = [t for t, u in zip((xr, na(xr), nb(xr)), use) if u]
        torch.autograd.backward(ref_outs, [g.float() for g in grads])
        want = [x.grad, na.weight.grad, na.bias.grad] + ([nb.weight.grad, nb.bias.grad] if use[2] else [])
        for g, w, nm, tol in zip(got, want, ('dx', 'dwa', 'dba', 'dwb', 'dbb'), (1e-4, 1e-3, 1e-3, 1e-3, 1e-3)):
            _close(g, w, tol, nm)
        if use[1]:
            _close(ya, na(x), 1e-2, 'ya')


@pytest.mark.parametrize('B,h,w,C,Co', [(2, 8, 16, 64, 32), (1, 5, 8, 128, 128), (2, 32, 32, 768, 768)])
def test_up_from_tokens_matches_conv_transpose(B, h, w, C, Co):
    """ConvTranspose2d(k 2, s 2) as GEMMs on token rows + sub-pixel interleave (fused.up_from_tokens; reference
    vit_adapter.py:46, 106-109) against F.conv_transpose2d in fp32 on the same bf16-rounded operands: output, d(rows),
    d(weight)."""
    import torch.nn.functional as F
    from vitadapter import fused
    torch.manual_seed(C + h)
    up = torch.nn.ConvTranspose2d(C, Co, 2, 2).cuda()
    rows = torch.randn(B, h * w, C, device='cuda', requires_grad=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        with fused.forward_epoch(up):
            out = fused.up_from_tokens(up, rows, h, w)
    assert out is not None and out.shape == (B, Co, 2 * h, 2 * w) and out.dtype == torch.bfloat16
    g = torch.randn_like(out, dtype=torch.float32)
    out.backward(g.to(torch.bfloat16))
    rr = rows.detach().to(torch.bfloat16).float().requires_grad_(True)
    wr = up.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    ref = F.conv_transpose2d(rr.transpose(1, 2).reshape(B, C, h, w), wr, None, stride=2)
    ref.backward(g.to(torch.bfloat16).float())
    _close(out.float(), ref, 2 ** -7, 'out')
    _close(rows.grad, rr.grad, 2 ** -6, 'd rows')
    _close(up.weight.grad, wr.grad, 2e-3, 'd weight')
    # with the addend of the tail (`up(c2) + c1`): one rounding of the sum, gradient passed through unchanged
    add = torch.randn(B, Co, 2 * h, 2 * w, device='cuda').to(torch.bfloat16).requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out2 = fused.up_from_tokens(up, rows.detach(), h, w, add)
    _close(out2.float(), ref.detach() + add.detach().float(), 2 ** -7, 'out + addend')
    out2.backward(g.to(torch.bfloat16))
    assert torch.equal(add.grad, g.to(torch.bfloat16))


@pytest.mark.parametrize('B,H,W,E', [(2, 64, 96, 192), (1, 224, 224, 768)])
def test_patch_embed_gemm_matches_conv(B, H, W, E):
    """Patch embedding (Conv2d, kernel = stride = 16, base/vit.py:169-190) as one GEMM on bf16 patch rows against the
    convolution in fp32 on the same bf16-rounded operands: tokens, d(weight), d(bias)."""
    from vitadapter import fused
    torch.manual_seed(E)
    conv = torch.nn.Conv2d(3, E, 16, 16).cuda()
    x = torch.randn(B, 3, H, W, device='cuda')
    with torch.autocast('cuda', dtype=torch.bfloat16):
        with fused.forward_epoch(conv):
            out = fused.patch_embed(conv, x)
    assert out is not None
    tok, Hp, Wp = out
    assert (Hp, Wp) == (H // 16, W // 16) and tok.shape == (B, Hp * Wp, E)
    g = torch.randn(tok.shape, device='cuda')
    tok.backward(g.to(tok.dtype))
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x.to(torch.bfloat16).float(), wr, br, stride=16).flatten(2).transpose(1, 2)
    ref.backward(g.to(torch.bfloat16).float())
    _close(tok.float(), ref, 2 ** -7, 'tokens')
    _close(conv.weight.grad, wr.grad, 2e-3, 'd weight')
    _close(conv.bias.grad, br.grad, 2e-3, 'd bias')


def test_drop_path_scales_are_pooled_per_forward():
    """vitadapter/fused.py::_DropPool: from the second forward of a module on, every drop-path site of the forward is
    served a row of ONE draw floor(keep + U) / keep (timm's DropPath formula) in the recorded order; values are 0 or
    1 / keep with the site's own keep; a site off the recorded sequence, a module that recomputes activations, eval mode
    and calls outside a forward epoch draw on their own."""
    from vitadapter import fused

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(64, 64)
            self.norm = torch.nn.LayerNorm(64)

        def forward(self, x, keeps, z):
            scales = []
            with fused.forward_epoch(self):
                for k in keeps:
                    dp = torch.nn.Dropout(0)            # any object with drop_prob / training
                    dp.drop_prob, dp.training = 1.0 - k, self.training
                    scales.append(fused._drop_path_scale(x, dp))
            return scales

    m = Toy().cuda().train()
    x = torch.randn(8, 16, 64, device='cuda')
    keeps = [0.9, 0.9, 0.7, 0.5]
    with torch.autocast('cuda', dtype=torch.bfloat16):
        first = m(x, keeps, None)
        assert fused.DROP_POOL.module is None and m.__dict__['_vah_drop_trace'] == [(k, 8) for k in keeps]
        assert all(s.data_ptr() != first[0].data_ptr() for s in first[1:])            # first forward: one draw per site
        seen = {k: set() for k in keeps}
        for _ in range(40):
            got = m(x, keeps, None)
            base = got[0].data_ptr()
            assert [s.data_ptr() - base for s in got] == [32 * i for i in range(4)]   # rows of one (4, 8) fp32 tensor
            for k, s in zip(keeps, got):
                vals = set(round(v, 5) for v in s.tolist())
                assert vals <= {0.0, round(1.0 / k, 5)}, (k, vals)
                seen[k] |= vals
        assert all(len(v) == 2 for k, v in seen.items() if k < 0.9) and seen[0.5] == {0.0, 2.0}
        off = m(x, [0.9, 0.6, 0.7, 0.5], None)                                        # second site differs: on-demand from there
        assert off[1].data_ptr() - off[0].data_ptr() != 32 and set(round(v, 4) for v in off[1].tolist()) <= {0.0, round(1 / 0.6, 4)}
        m.eval()
        assert m(x, keeps, None) == [None] * 4
        m.train()
        m.with_cp = True                                                              # a module that recomputes: never pooled
        m.__dict__.pop('vah_drop_pool_ok')
        a = m(x, keeps, None)
        b = m(x, keeps, None)
        assert b[1].data_ptr() - b[0].data_ptr() != 32 and a is not b
    dp = torch.nn.Dropout(0)
    dp.drop_prob, dp.training = 0.5, True
    assert fused._drop_path_scale(x, dp).shape == (8,)                                # outside an epoch: plain draw
