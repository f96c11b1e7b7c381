"""GPU: the implicit-GEMM 3x3 convolutions (csrc/conv.hip) against torch's convolution in fp32 on the same bf16
operands - forward, input gradient (stride 1 and, through the four output parities, stride 2) and weight gradient
of every conv of the SpatialPriorModule (adapter_modules.py:217-260), at small spatial sizes with ragged edges.
Tolerance: bf16 output rounding (2^-8 relative to the largest value) for the bf16 outputs, 1e-3 for the fp32 dW."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(16, 64, 2), (64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 256, 2), (128, 64, 1)]


def _inputs(cin, cout, N, H, W, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(N, H, W, cin, device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * (9 * cin) ** -0.5).to(torch.bfloat16)
    return x, w


@pytest.mark.parametrize('cin,cout,stride', CASES)
@pytest.mark.parametrize('hw', [(8, 32), (13, 45), (66, 70)])
def test_conv_forward(cin, cout, stride, hw):
    from vitadapter import conv
    x, w = _inputs(cin, cout, 2, hw[0], hw[1], 1)
    got = conv.conv3x3_forward(x, conv.forward_weight(w), stride)
    want = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, stride, 1).permute(0, 2, 3, 1)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert (got.float() - want).abs().max().item() <= 2 ** -7 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('cin,cout,stride', CASES)
@pytest.mark.parametrize('hw', [(8, 32), (13, 45), (66, 70)])
def test_conv_input_gradient(cin, cout, stride, hw):
    from vitadapter import conv
    x, w = _inputs(cin, cout, 2, hw[0], hw[1], 2)
    if cin == 16:
        pytest.skip('the image needs no gradient: the 16-channel stem has no input-gradient path')
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    y = F.conv2d(xr, w.float(), None, stride, 1)
    gy = torch.randn_like(y).to(torch.bfloat16)
    want = torch.autograd.grad(y, xr, gy.float())[0].permute(0, 2, 3, 1)
    got = conv.conv3x3_input_grad(gy.permute(0, 2, 3, 1).contiguous(), conv.dgrad_weight(w), stride, hw)
    assert got.shape == want.shape
    assert (got.float() - want).abs().max().item() <= 2 ** -7 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('cin,cout,stride', CASES)
@pytest.mark.parametrize('hw', [(8, 32), (13, 45), (66, 70)])
def test_conv_weight_gradient(cin, cout, stride, hw):
    from vitadapter import conv
    x, w = _inputs(cin, cout, 2, hw[0], hw[1], 3)
    wr = w.float().requires_grad_(True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), wr, None, stride, 1)
    gy = torch.randn_like(y).to(torch.bfloat16)
    want = torch.autograd.grad(y, wr, gy.float())[0]
    got = conv.conv3x3_weight_grad(x, gy.permute(0, 2, 3, 1).contiguous(), stride)          # (cout, 3, 3, cin) fp32
    got = got.permute(0, 3, 1, 2)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert (got - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())
    again = conv.conv3x3_weight_grad(x, gy.permute(0, 2, 3, 1).contiguous(), stride).permute(0, 3, 1, 2)
    assert torch.equal(got, again)                       # fixed summation order
