#!/usr/bin/env python
"""A few launches of the plain fp32 MSDeformAttnFunction kernels (the 8b boundary) at a BASELINE call
shape, for rocprofv3 runs.  Usage: prof_msda_plain.py cfg3_ext [iters] [adapter|uniform]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import MultiScaleDeformableAttention as MSDA  # noqa: E402
from test_msda_gpu import _full_inputs  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg3_ext'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else 'adapter'
v, s, i, l, a, g = _full_inputs(cfg, mode)
for _ in range(iters):
    out = MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
torch.cuda.synchronize()
print('done', cfg, mode, float(out.abs().mean()), float(gv.abs().mean()))
