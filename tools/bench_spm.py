#!/usr/bin/env python
"""Spatial prior module (base_det: 2 x 3 x 1024 x 1024, inplanes 64, embed 768) under bf16 autocast: the whole module
forward + backward, and each 3x3 convolution alone (forward, backward) in NCHW and channels_last."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from bench_msda import timeit  # noqa: E402
from vitadapter.backbones.adapter_modules import SpatialPriorModule  # noqa: E402


def main():
    torch.manual_seed(0)
    spm = SpatialPriorModule(64, 768).cuda().train()
    x = torch.randn(2, 3, 1024, 1024, device='cuda')

    def step():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            outs = spm(x)
        sum(o.float().sum() for o in outs).backward()
    print('SPM fwd+bwd %.1f us' % (timeit(step, iters=10, warm=5) * 1e6))
    import _vah
    from vitadapter import fused, spm_nhwc
    level = torch.zeros(3, 768, device='cuda', requires_grad=True)

    def step_nhwc():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            with fused.forward_epoch(spm):
                c1, c = spm_nhwc.forward(spm, x, level)
        (c1.float().sum() + c.sum()).backward()
    step_nhwc()
    _vah.prof_enable(True, 'spm_,conv_,gemm_,colsum')
    print('SPM NHWC fwd+bwd %.1f us' % (timeit(step_nhwc, iters=10, warm=3) * 1e6))
    rep = _vah.prof_report()
    _vah.prof_enable(False)
    for name, row in sorted(rep.items(), key=lambda kv: -kv[1]['total_ms']):
        print('   %-22s calls/step %5.1f  avg %7.1f us  %7.1f us/step' % (name, row['calls'] / 13, row['total_ms'] / row['calls'] * 1e3, row['total_ms'] / 13 * 1e3))
    from vitadapter import conv
    for cin, cout, hw, stride in [(16, 64, 1024, 2), (64, 64, 512, 1), (64, 128, 256, 2), (128, 256, 128, 2), (256, 256, 64, 2)]:
        xi = torch.randn(2, hw, hw, cin, device='cuda').to(torch.bfloat16)
        w = torch.randn(cout, cin, 3, 3, device='cuda').to(torch.bfloat16)
        w9 = conv.forward_weight(w)
        wt9 = conv.dgrad_weight(w)
        y = conv.conv3x3_forward(xi, w9, stride)
        g = torch.randn_like(y)
        tf = timeit(lambda: conv.conv3x3_forward(xi, w9, stride), iters=10, warm=3)
        td = timeit(lambda: conv.conv3x3_input_grad(g, wt9, stride, (hw, hw)), iters=10, warm=3) if cin != 16 else 0.
        tw = timeit(lambda: conv.conv3x3_weight_grad(xi, g, stride), iters=10, warm=3)
        fl = 2 * 2 * cout * cin * 9 * (hw // stride) ** 2
        print('own  %3d->%3d @%4d s%d nhwc           fwd %7.1f us (%5.1f TF/s)  dgrad %7.1f us  wgrad %7.1f us (%5.1f TF/s)'
              % (cin, cout, hw, stride, tf * 1e6, fl / tf / 1e12, td * 1e6, tw * 1e6, fl / tw / 1e12), flush=True)
    shapes = [(3, 64, 1024, 2), (64, 64, 512, 1), (64, 64, 512, 1), (64, 128, 256, 2), (128, 256, 128, 2), (256, 256, 64, 2)]
    for cin, cout, hw, stride in shapes:
        for fmt in (torch.contiguous_format, torch.channels_last):
            xi = torch.randn(2, cin, hw, hw, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=fmt).requires_grad_(True)
            w = torch.randn(cout, cin, 3, 3, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=fmt).requires_grad_(True)
            y = F.conv2d(xi, w, None, stride, 1)
            g = torch.randn_like(y)
            tf = timeit(lambda: F.conv2d(xi, w, None, stride, 1), iters=10, warm=5)
            tb = timeit(lambda: torch.autograd.grad(y, [xi, w], g, retain_graph=True), iters=10, warm=5)
            fl = 2 * 2 * cout * cin * 9 * (hw // stride) ** 2
            print('conv %3d->%3d @%4d s%d %-14s fwd %7.1f us (%5.1f TF/s)  bwd %7.1f us (%5.1f TF/s)'
                  % (cin, cout, hw, stride, 'channels_last' if fmt == torch.channels_last else 'nchw', tf * 1e6, fl / tf / 1e12,
                     tb * 1e6, 2 * fl / tb / 1e12), flush=True)


if __name__ == '__main__':
    main()
