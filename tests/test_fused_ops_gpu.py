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
