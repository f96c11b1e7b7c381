#!/usr/bin/env python
"""Where do the tiny torch kernels of a training step come from?  One bench-shaped step under
torch.profiler with stacks; prints call counts per (op, python frame) for the given aten ops."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from vitadapter.backbones.vit_adapter import PRESETS, ViTAdapter  # noqa: E402


def main():
    ops = sys.argv[1:] or ['aten::fill_', 'aten::zero_', 'aten::lt', 'aten::gt', 'aten::le', 'aten::ge', 'aten::eq',
                           'aten::ne', 'aten::copy_', 'aten::zeros', 'aten::zeros_like', 'aten::ones_like']
    dev = torch.device('cuda')
    torch.manual_seed(0)
    model = ViTAdapter(**PRESETS['base_det']).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.05, fused=True)
    x = torch.randn(2, 3, 1024, 1024, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            feats = model(x)
        loss = sum(f.float().mean() for f in feats)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
    torch.cuda.synchronize()
    counts = Counter()
    for ev in prof.events():
        if ev.name in ops:
            frames = [f for f in (ev.stack or []) if 'vit-adapter_amd' in f or 'bench' in f or 'optim' in f]
            where = frames[0] if frames else ((ev.stack or ['<autograd / no python frame>'])[0])
            counts[(ev.name, where.strip()[-110:])] += 1
    for (name, where), n in counts.most_common(60):
        print('%5d  %-18s %s' % (n, name, where))


if __name__ == '__main__':
    main()
