#!/usr/bin/env python
"""Which Python lines issue the small fill / zero kernels of one eager training step (torch profiler with stacks)."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from vitadapter.backbones.vit_adapter import PRESETS, ViTAdapter  # noqa: E402

torch.manual_seed(0)
model = ViTAdapter(**dict(PRESETS['base_det'])).cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5, fused=True)
x = torch.randn(2, 3, 1024, 1024, device='cuda')


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        feats = model(x)
    sum(f.float().mean() for f in feats).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
names = ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::zeros_like', 'aten::ones', 'aten::full', 'aten::new_zeros')
sites = Counter()
for e in prof.events():
    if e.name in names:
        st = [s for s in e.stack if 'vit-adapter_amd' in s or 'bench' in s or 'autograd' in s]
        sites[(e.name, st[0] if st else (e.stack[0] if e.stack else '?'))] += 1
for (n, s), c in sites.most_common(30):
    print('%4d  %-18s %s' % (c, n, s[-110:]))
