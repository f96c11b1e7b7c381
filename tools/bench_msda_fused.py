#!/usr/bin/env python
"""Micro-benchmark of the fused MSDA core (forward; backward through the tile pass and, for A/B, through
per-sample atomics) at the BASELINE call shapes, bf16 IO as under autocast.  HIP-event timed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench_msda import timeit  # noqa: E402
from oracle import cases  # noqa: E402
from ops.functions import MSDeformAttnFusedFunction  # noqa: E402


def main():
    cfgs = sys.argv[1:] or ['cfg3_inj', 'cfg3_ext']
    dt = torch.bfloat16
    for cfg in cfgs:
        N, M, D, P, Lq, shapes, qshapes = cases.bench_inputs(cfg)
        L, S = len(shapes), sum(h * w for h, w in shapes)
        g = torch.Generator(device='cuda').manual_seed(0)
        value = torch.randn(N, S, M, D, device='cuda', generator=g).to(dt).requires_grad_(True)
        off = (cases.ring_offsets(M, L, P).cuda()[None, None] + float(os.environ.get("VAH_BENCH_NOISE", "1")) * torch.randn(N, Lq, M, L, P, 2, device="cuda", generator=g)).to(dt).requires_grad_(True)
        logit = torch.randn(N, Lq, M, L * P, device='cuda', generator=g).to(dt).requires_grad_(True)
        ref = cases.reference_grid(qshapes).cuda()
        hw = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
        lsi = cases.level_start_index(shapes).cuda()
        gout = torch.randn(N, Lq, M * D, device='cuda', generator=g).to(dt)
        mb_f = 2 * (N * S * M * D + N * Lq * M * D) + 2 * 3 * N * Lq * M * L * P          # bytes moved with bf16 IO
        mb_b = 2 * (2 * N * S * M * D + N * Lq * M * D) + 2 * 6 * N * Lq * M * L * P
        for tiled in ('1', '0'):
            os.environ['VAH_MSDA_TILED'] = tiled
            out = MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref)
            tf = timeit(lambda: MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref))

            def bwd():
                torch.autograd.grad(out, [value, off, logit], gout, retain_graph=True)
            tb = timeit(bwd)
            print('%-9s tiled=%s fwd %7.1f us (%.3f of 8 TB/s on moved bytes) | bwd %8.1f us (%.3f)'
                  % (cfg, tiled, tf * 1e6, mb_f / tf / 8e12, tb * 1e6, mb_b / tb / 8e12), flush=True)


if __name__ == '__main__':
    main()
