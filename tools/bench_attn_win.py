#!/usr/bin/env python
"""Micro-benchmark of the window attention kernels at the base_det shape (2 x 64 x 64 tokens, 12 heads, 14 x 14 windows)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench_msda import timeit  # noqa: E402
from vitadapter import kernels  # noqa: E402


def main():
    B, gh, gw, H, win = 2, 64, 64, 12, 14
    if len(sys.argv) > 1:
        B, gh, gw, H, win = [int(a) for a in sys.argv[1:6]]
    qkv = torch.randn(B, gh * gw, 3, H, 64, device='cuda').to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(B, gh * gw, H, 64, device='cuda').to(torch.bfloat16)
    out = kernels.window_attention(qkv, 0.125, gh, gw, win)
    tf = timeit(lambda: kernels.window_attention(qkv, 0.125, gh, gw, win), iters=50)
    tb = timeit(lambda: torch.autograd.grad(out, qkv, g, retain_graph=True), iters=50)
    Z = B * (-(-gh // win)) * (-(-gw // win))
    fl = 4 * Z * H * win ** 4 * 64
    print('win attention %dx%dx%d heads %d win %d: fwd %.1f us (%.3f of 2.5 PF) | bwd %.1f us (%.3f)'
          % (B, gh, gw, H, win, tf * 1e6, fl / tf / 2.5e15, tb * 1e6, 2.5 * fl / tb / 2.5e15))


if __name__ == '__main__':
    main()
