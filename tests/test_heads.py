"""mmseg's UPerHead / FCNHead restated (vitadapter/heads.py, SURVEY section 8 f-4, BASELINE configs[3]).  mmseg is not in
the reference tree, which only configures the heads (configs/_base_/models/upernet_r50.py:17-41): PARITY UNPINNED against
mmseg; the modules are held to a functional re-evaluation of the published arithmetic from their state_dict, and their
parameter names to mmseg's checkpoint layout."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn


def _cm(sd, prefix, x, pad):
    x = F.conv2d(x, sd[prefix + '.conv.weight'], None, padding=pad)
    x = F.batch_norm(x, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'], sd[prefix + '.bn.weight'],
                     sd[prefix + '.bn.bias'], False, 0.0, 1e-5)
    return F.relu(x)


def _uper_expected(sd, feats, scales):
    up = lambda t, size: F.interpolate(t, size=size, mode='bilinear', align_corners=False)      # noqa: E731
    x = feats[-1]
    psp = [x] + [up(_cm(sd, 'psp_modules.%d.1' % i, F.adaptive_avg_pool2d(x, s), 0), x.shape[2:]) for i, s in enumerate(scales)]
    lat = [_cm(sd, 'lateral_convs.%d' % i, feats[i], 0) for i in range(3)] + [_cm(sd, 'bottleneck', torch.cat(psp, 1), 1)]
    for i in range(3, 0, -1):
        lat[i - 1] = lat[i - 1] + up(lat[i], lat[i - 1].shape[2:])
    outs = [_cm(sd, 'fpn_convs.%d' % i, lat[i], 1) for i in range(3)] + [lat[3]]
    outs = [outs[0]] + [up(o, outs[0].shape[2:]) for o in outs[1:]]
    y = _cm(sd, 'fpn_bottleneck', torch.cat(outs, 1), 1)
    return F.conv2d(y, sd['conv_seg.weight'], sd['conv_seg.bias'])


def _feats(dev, C=(8, 12, 16, 24), hw=(24, 32), batch=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, c, hw[0] >> i, hw[1] >> i, generator=g).to(dev) for i, c in enumerate(C)]


def _heads(dev, norm=nn.BatchNorm2d):
    from vitadapter.heads import FCNHead, UPerHead
    torch.manual_seed(2)
    u = UPerHead(in_channels=(8, 12, 16, 24), channels=16, num_classes=7, norm=norm)
    f = FCNHead(in_channels=16, in_index=2, channels=8, num_classes=7, norm=norm)
    for m in list(u.modules()) + list(f.modules()):
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            with torch.no_grad():
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.normal_(1, 0.1)
                m.bias.normal_(0, 0.1)
    return u.to(dev), f.to(dev)


def test_state_dict_follows_mmseg_layout():
    from vitadapter.heads import FCNHead, UPerHead
    u = UPerHead()                                       # configs[3] defaults: 4 x 1024 channels in, 512 inside, 150 classes
    keys = set(u.state_dict())
    for k in ('psp_modules.0.1.conv.weight', 'psp_modules.3.1.bn.running_var', 'bottleneck.conv.weight', 'lateral_convs.2.bn.bias',
              'fpn_convs.0.conv.weight', 'fpn_bottleneck.bn.weight', 'conv_seg.weight', 'conv_seg.bias'):
        assert k in keys, k
    assert u.bottleneck.conv.weight.shape == (512, 1024 + 4 * 512, 3, 3) and u.fpn_bottleneck.conv.weight.shape == (512, 2048, 3, 3)
    assert u.conv_seg.weight.shape == (150, 512, 1, 1) and u.bottleneck.conv.bias is None
    assert isinstance(u.bottleneck.bn, nn.SyncBatchNorm)
    f = FCNHead()
    assert set(f.state_dict()) >= {'convs.0.conv.weight', 'convs.0.bn.weight', 'conv_seg.weight'} and f.convs[0].conv.weight.shape == (256, 1024, 3, 3)


def test_heads_equal_functional_evaluation_cpu():
    u, f = _heads('cpu')
    u.eval(), f.eval()
    feats = _feats('cpu')
    with torch.no_grad():
        out = u(feats)
        want = _uper_expected(u.state_dict(), feats, (1, 2, 3, 6))
        assert out.shape == (2, 7, 24, 32) and (out - want).abs().max().item() <= 1e-5
        sd = f.state_dict()
        want_f = F.conv2d(_cm(sd, 'convs.0', feats[2], 1), sd['conv_seg.weight'], sd['conv_seg.bias'])
        assert (f(feats) - want_f).abs().max().item() <= 1e-5
    u.train()                                            # dropout + batch statistics: runs, differentiable
    u(feats).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in u.parameters())


@pytest.mark.gpu
def test_heads_on_gpu_fp32_and_bf16():
    u, f = _heads('cuda')
    u.eval(), f.eval()
    feats = _feats('cuda')
    with torch.no_grad():
        out = u(feats)
        want = _uper_expected({k: v.cpu() for k, v in u.state_dict().items()}, [t.cpu() for t in feats], (1, 2, 3, 6))
        assert (out.cpu() - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out16 = u(feats)
        assert float((out16.float() - out).norm() / out.norm()) <= 3e-2
    u.train()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = u(feats).float().mean() + f(feats).float().mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(u.parameters()) + list(f.parameters()))
