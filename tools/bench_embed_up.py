#!/usr/bin/env python
"""The two convolutions of base_det that still go through MIOpen: patch embedding (16 x 16, stride 16, 3 -> 768, on
2 x 3 x 1024 x 1024) and `up` (ConvTranspose2d 768 -> 768, k 2, s 2, on 2 x 768 x 128 x 128), bf16 autocast, GPU time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from bench_msda import timeit  # noqa: E402


def main():
    x = torch.randn(2, 3, 1024, 1024, device='cuda')
    pe = torch.nn.Conv2d(3, 768, 16, 16).cuda()
    up = torch.nn.ConvTranspose2d(768, 768, 2, 2).cuda()
    c2 = torch.randn(2, 768, 128, 128, device='cuda', requires_grad=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = pe(x)
        u = F.conv_transpose2d(c2, up.weight, None, stride=2)
        tf1 = timeit(lambda: pe(x), iters=10, warm=5)
        tf2 = timeit(lambda: F.conv_transpose2d(c2, up.weight, None, stride=2), iters=10, warm=5)
    g1, g2 = torch.randn_like(y), torch.randn_like(u)
    tb1 = timeit(lambda: torch.autograd.grad(y, [pe.weight, pe.bias], g1, retain_graph=True), iters=10, warm=5)
    tb2 = timeit(lambda: torch.autograd.grad(u, [c2, up.weight], g2, retain_graph=True), iters=10, warm=5)
    print('patch_embed fwd %.1f us  bwd %.1f us | up fwd %.1f us  bwd %.1f us' % (tf1 * 1e6, tb1 * 1e6, tf2 * 1e6, tb2 * 1e6))


if __name__ == '__main__':
    main()
