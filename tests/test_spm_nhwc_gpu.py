"""GPU: the SpatialPriorModule on NHWC bf16 with the implicit-GEMM convolutions (vitadapter/spm_nhwc.py) against the
same module run as the reference writes it (NCHW, torch convolutions, adapter_modules.py:217-268) in FP32 on the same
weights: outputs and every parameter gradient, at bf16 tolerances (operands rounded to 8 bits at six convolutions and
six BatchNorms); and the NHWC BatchNorm / max-pool kernels alone against torch."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_bn_relu_nhwc_matches_torch():
    from vitadapter import spm_nhwc
    torch.manual_seed(0)
    for C, shape in ((64, (2, 24, 40)), (256, (3, 7, 9)), (128, (1, 16, 16))):
        x = (torch.randn(*shape, C, device='cuda') * 2 + 0.5).to(torch.bfloat16).requires_grad_(True)
        norm = torch.nn.BatchNorm2d(C).cuda().train()
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5)
            norm.bias.uniform_(-0.5, 0.5)
        ref = torch.nn.BatchNorm2d(C).cuda().train()
        ref.load_state_dict(norm.state_dict())
        y = spm_nhwc._BNRelu.apply(x, norm.weight, norm.bias, norm, True)
        g = torch.randn_like(y)
        y.backward(g)
        xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = F.relu(ref(xr))
        yr.backward(g.float().permute(0, 3, 1, 2))
        assert (y.float() - yr.permute(0, 2, 3, 1)).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
        assert (x.grad.float() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() <= 2 ** -6 * max(1.0, xr.grad.abs().max().item())
        assert torch.allclose(norm.weight.grad, ref.weight.grad, rtol=2e-2, atol=2e-2 * ref.weight.grad.abs().max().item())
        assert torch.allclose(norm.bias.grad, ref.bias.grad, rtol=2e-2, atol=2e-2 * ref.bias.grad.abs().max().item())
        assert torch.allclose(norm.running_mean, ref.running_mean, atol=1e-4)
        assert torch.allclose(norm.running_var, ref.running_var, rtol=1e-3, atol=1e-4)


def test_maxpool_nhwc_matches_torch():
    from vitadapter import spm_nhwc
    torch.manual_seed(1)
    for shape in ((2, 16, 24, 64), (1, 7, 9, 8), (2, 33, 32, 16)):
        x = torch.randn(*shape, device='cuda').to(torch.bfloat16).requires_grad_(True)
        y = spm_nhwc._MaxPool.apply(x)
        g = torch.randn_like(y)
        y.backward(g)
        xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = F.max_pool2d(xr, 3, 2, 1)
        yr.backward(g.float().permute(0, 3, 1, 2))
        assert torch.equal(y.float(), yr.permute(0, 2, 3, 1))
        assert (x.grad.float() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() <= 2 ** -7 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize('hw', [(160, 224), (256, 256)])
def test_spm_nhwc_matches_reference_order(hw):
    from vitadapter import fused, spm_nhwc
    from vitadapter.backbones.adapter_modules import SpatialPriorModule
    torch.manual_seed(2)
    spm = SpatialPriorModule(64, 128).cuda().train()
    ref = SpatialPriorModule(64, 128).cuda().train()
    ref.load_state_dict(spm.state_dict())
    level = torch.randn(3, 128, device='cuda', requires_grad=True)
    level_r = level.detach().clone().requires_grad_(True)
    x = torch.randn(2, 3, *hw, device='cuda')
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert spm_nhwc.usable(spm, x)
        with fused.forward_epoch(spm):
            c1, c = spm_nhwc.forward(spm, x, level)
    # the reference order, fp32 throughout (no fused ops: autocast off)
    r1, r2, r3, r4 = ref(x)
    rc = torch.cat([r2 + level_r[0], r3 + level_r[1], r4 + level_r[2]], dim=1)
    r1 = r1 - ref.fc1.bias.view(1, -1, 1, 1)                     # the NHWC path returns c1 without fc1's bias
    g1, gc = torch.randn_like(r1), torch.randn_like(rc)
    ((c1.float() * g1).sum() + (c * gc).sum()).backward()
    ((r1 * g1).sum() + (rc * gc).sum()).backward()

    def rel(a, b):
        return float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-12))
    assert c1.shape == r1.shape and c.shape == rc.shape and c.dtype == torch.float32
    assert rel(c1, r1) <= 3e-2 and rel(c, rc) <= 3e-2, (rel(c1, r1), rel(c, rc))
    assert rel(level.grad, level_r.grad) <= 3e-2
    # yardstick for the gradients: the SAME module on the NCHW bf16 autocast path (torch convolutions, the NCHW
    # BatchNorm / max-pool kernels) against the same fp32 run.  ReLU masks and max-pool winners flip under 8-bit
    # operands, so both bf16 paths sit 0.1 - 0.2 (relative L2) from fp32 on the convolution weights; the NHWC path
    # must be as close to fp32 as that path is (2x + 3e-2 slack: two noisy estimates are compared), parameter by
    # parameter.  Maps of at least 5 x 7 at stride 32: below that the last BatchNorm averages over a few dozen samples.
    old = SpatialPriorModule(64, 128).cuda().train()
    old.load_state_dict(ref.state_dict())
    for m, r in zip(old.modules(), ref.modules()):
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.zero_(), m.running_var.fill_(1.), m.num_batches_tracked.zero_()
    level_o = level.detach().clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        o1, o2, o3, o4 = old(x)
        oc = torch.cat([o2 + level_o[0], o3 + level_o[1], o4 + level_o[2]], dim=1)
        o1 = o1 - old.fc1.bias.view(1, -1, 1, 1)
    ((o1.float() * g1).sum() + (oc.float() * gc).sum()).backward()
    worst = {}
    for (k, p), (_, q), (_, o) in zip(spm.named_parameters(), ref.named_parameters(), old.named_parameters()):
        if k == 'fc1.bias' or q.grad is None or float(q.grad.norm()) == 0.:
            continue
        worst[k] = (rel(p.grad, q.grad), rel(o.grad, q.grad))
    bad = {k: v for k, v in worst.items() if v[0] > 2.0 * v[1] + 3e-2}
    assert len(worst) >= 20 and not bad and max(v[0] for v in worst.values()) <= 0.3, (bad, sorted(worst.items(), key=lambda kv: -kv[1][0])[:4])
    for a, b in zip(spm.buffers(), ref.buffers()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, b, rtol=2e-2, atol=2e-3)
