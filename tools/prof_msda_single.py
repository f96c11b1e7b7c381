#!/usr/bin/env python
"""A few launches of the fused MSDA core (forward + backward, bf16 IO as in bench.py) at a BASELINE
call shape, for rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE per launch -> roofline.traffic).
Usage: prof_msda_single.py cfg3_ext [iters] [offset noise in px: 1 = "adapter" offsets, 0 = the ring bias of a
freshly initialised model, which is what bench.py runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle import cases  # noqa: E402
from ops.functions import MSDeformAttnFusedFunction  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3_ext'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dt = torch.bfloat16
N, M, D, P, Lq, shapes, qshapes = cases.bench_inputs(cfg)
L, S = len(shapes), sum(h * w for h, w in shapes)
g = torch.Generator(device='cuda').manual_seed(0)
value = torch.randn(N, S, M, D, device='cuda', generator=g).to(dt).requires_grad_(True)
off = (cases.ring_offsets(M, L, P).cuda()[None, None]
       + noise * torch.randn(N, Lq, M, L, P, 2, device='cuda', generator=g)).to(dt).requires_grad_(True)
logit = torch.randn(N, Lq, M, L * P, device='cuda', generator=g).to(dt).requires_grad_(True)
ref = cases.reference_grid(qshapes).cuda()
hw = torch.as_tensor(shapes, dtype=torch.long, device='cuda')
lsi = cases.level_start_index(shapes).cuda()
gout = torch.randn(N, Lq, M * D, device='cuda', generator=g).to(dt)
for _ in range(iters):
    out = MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref)
    torch.autograd.grad(out, [value, off, logit], gout)
torch.cuda.synchronize()
print('done', cfg, float(out.float().abs().mean()))
