#!/usr/bin/env python
"""Row-streaming kernels of the blocks alone: residual + LayerNorm forward / backward at the two row counts of base_det
(2 x 4096 ViT tokens, 2 x 21504 adapter tokens), C = 768, with the bytes each pass moves."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench_msda import timeit  # noqa: E402
from vitadapter import fused  # noqa: E402


def main():
    C = 768
    norm = torch.nn.LayerNorm(C, eps=1e-6).cuda()
    gamma = torch.ones(C, device='cuda', requires_grad=True)
    import _vah
    _vah.prof_enable(True, 'layernorm,residual_layernorm')
    for rows in (8192, 43008):
        x = torch.randn(2, rows // 2, C, device='cuda', requires_grad=True)
        z = torch.randn(2, rows // 2, C, device='cuda').to(torch.bfloat16).requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            t, y = fused.residual_ln(x, z, gamma, None, norm)
            tf = timeit(lambda: fused.residual_ln(x, z, gamma, None, norm), iters=30)
        gy, gt = torch.randn_like(y), torch.randn_like(t)
        tb = timeit(lambda: torch.autograd.grad([t, y], [x, z, gamma, norm.weight, norm.bias], [gt, gy], retain_graph=True), iters=30)
        n = rows * C
        print('rows %6d residual+LN fwd %6.1f us (%.2f TB/s on %d MB) | bwd %6.1f us (%.2f TB/s on %d MB)'
              % (rows, tf * 1e6, n * 12 / tf / 1e12, n * 12 >> 20, tb * 1e6, n * 18 / tb / 1e12, n * 18 >> 20), flush=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y2 = fused.layer_norm(norm, x)
            tf = timeit(lambda: fused.layer_norm(norm, x), iters=30)
        g2 = torch.randn_like(y2)
        tb = timeit(lambda: torch.autograd.grad(y2, [x, norm.weight, norm.bias], g2, retain_graph=True), iters=30)
        rep = _vah.prof_report()
        _vah.prof_enable(True, 'layernorm,residual_layernorm')
        print('   GPU time per launch: ' + ', '.join('%s %.1f us' % (k, v['total_ms'] / v['calls'] * 1e3) for k, v in sorted(rep.items())))
        print('rows %6d LN          fwd %6.1f us (%.2f TB/s on %d MB) | bwd %6.1f us (%.2f TB/s on %d MB)'
              % (rows, tf * 1e6, n * 6 / tf / 1e12, n * 6 >> 20, tb * 1e6, n * 10 / tb / 1e12, n * 10 >> 20), flush=True)


if __name__ == '__main__':
    main()
