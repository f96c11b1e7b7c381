#!/usr/bin/env python
"""Micro-benchmark of the fused MSDA core (forward, backward with / without the pull schedule) at
the BASELINE call shapes, bf16 IO as under autocast.  HIP-event timed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench_msda import timeit  # noqa: E402
from oracle import cases  # noqa: E402
from ops.functions import MSDeformAttnFusedFunction  # noqa: E402
from ops.functions import ms_deform_attn_fused as mf  # noqa: E402


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
        fb = 4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D)
        bb = 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D)
        for pull in ('0', '1'):
            os.environ['VAH_MSDA_PULL'] = pull
            mf._PULL_CACHE.clear()
            out = MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref)
            tf = timeit(lambda: MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref))

            def bwd():
                torch.autograd.grad(out, [value, off, logit], gout, retain_graph=True)
            tb = timeit(bwd)
            sched = mf.pull_schedule_for(ref, hw)
            extra = '' if sched is None else ' tiles %d cand %d' % (sched.ntiles, sched.cand.numel())
            print('%-9s pull=%s fwd %7.1f us (%.3f) | bwd(+zero-fill,+cast) %8.1f us (%.3f)%s'
                  % (cfg, pull, tf * 1e6, fb / tf / 8e12, tb * 1e6, bb / tb / 8e12, extra), flush=True)


if __name__ == '__main__':
    main()
