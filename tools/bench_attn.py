#!/usr/bin/env python
"""Micro-benchmark of the global attention kernels at the base_det shape (2 x 4096 tokens, 12 heads of 64)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _vah  # noqa: E402
from bench_msda import timeit  # noqa: E402
from vitadapter import kernels  # noqa: E402


def main():
    B, N, H = 2, 4096, 12
    if len(sys.argv) > 1:
        B, N, H = [int(a) for a in sys.argv[1:4]]
    qkv = torch.randn(B, N, 3, H, 64, device='cuda').to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(B, N, H, 64, device='cuda').to(torch.bfloat16)
    out = kernels.attention(qkv, 0.125)
    _vah.prof_enable(True, 'attn_')
    tf = timeit(lambda: kernels.attention(qkv, 0.125), iters=30)
    tb = timeit(lambda: torch.autograd.grad(out, qkv, g, retain_graph=True), iters=30)
    fl = 4 * B * H * N * N * 64
    print('attention %dx%d heads %d: fwd %.1f us (%.3f of 2.5 PF) | bwd %.1f us (%.3f)'
          % (B, N, H, tf * 1e6, fl / tf / 2.5e15, tb * 1e6, 2.5 * fl / tb / 2.5e15))
    for name, row in sorted(_vah.prof_report().items()):
        us = row['total_ms'] / row['calls'] * 1e3
        print('  %-24s %7.1f us  %.3f of MFMA peak' % (name, us, row['flops'] / row['calls'] / (us * 1e-6) / 2.5e15))


if __name__ == '__main__':
    main()
