#!/usr/bin/env python
"""A few launches of the fused MSDA core (forward + backward, bf16 IO as in bench.py) at a BASELINE
call shape, for rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE per launch -> roofline.traffic).
Usage: prof_msda_single.py cfg3_ext [iters] [offset noise in px: 1 = "adapter" offsets, 0 = the ring bias of a
freshly initialised model, which is what bench.py runs] [form: pair (default) = the call the module makes since round 3 -
fp32 [offsets | logits] rows of the pair GEMM read in place, bf16 gradient rows - or split = separate bf16 tensors]"""
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
form = sys.argv[4] if len(sys.argv) > 4 else 'pair'
if form == 'split':
    for _ in range(iters):
        out = MSDeformAttnFusedFunction.apply(value, hw, lsi, off, logit, ref)
        torch.autograd.grad(out, [value, off, logit], gout)
else:
    import _vah  # noqa: E402
    from ops.functions import ms_deform_attn_fused as mf  # noqa: E402
    PS = 3 * L * P
    y = torch.cat((off.detach().float().reshape(N, Lq, M, 2 * L * P), logit.detach().float()), -1).contiguous()   # (N, Lq, M, PS)
    offs, logs = y[..., :2 * L * P].unflatten(-1, (L, P, 2)), y[..., 2 * L * P:]
    refc = ref.float().contiguous().view(Lq, -1, 2)
    v = value.detach()
    for _ in range(iters):
        out = mf.fused_forward(v, hw, lsi, offs, logs, PS, PS, refc)
        gm = torch.empty((N * Lq, M * PS), dtype=torch.bfloat16, device='cuda')
        gv = torch.empty_like(v)
        ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
        with _vah.on(v.device):
            _vah.check(_vah.lib.vah_msda_fused_backward_tiled(
                v.data_ptr(), 1, hw.data_ptr(), lsi.data_ptr(), y.data_ptr(), y.data_ptr() + 2 * L * P * 4, 0, PS, PS,
                refc.data_ptr(), refc.shape[1], gout.data_ptr(), N, S, M, D, L, Lq, P, gv.data_ptr(), 1,
                gm.data_ptr(), gm.data_ptr() + 2 * L * P * 2, 1, PS, PS, ws.data_ptr(), ws_bytes, _vah.raw_stream(v.device)),
                'vah_msda_fused_backward_tiled')
torch.cuda.synchronize()
print('done', cfg, float(out.float().abs().mean()))
