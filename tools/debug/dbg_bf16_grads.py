"""Which parameter gradients differ most between the fp32 and the bf16-autocast run of the small seg backbone?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
    sys.path.insert(0, p)
import torch
from oracle import backbone_cases as bc, seeded
from vitadapter.backbones import ViTAdapter

name = sys.argv[1] if len(sys.argv) > 1 else 'seg_glob_64'
case = bc.FULL_CASES[name]
model = ViTAdapter(**case['cfg'])
shapes = {k: tuple(t.shape) for k, t in model.state_dict().items()}
model.load_state_dict(seeded.seeded_state_dict(shapes, 5))
model = model.cuda().train()
x = bc.full_input(name).cuda()
res = {}
for amp in (False, True):
    model.load_state_dict(seeded.seeded_state_dict(shapes, 5))
    model.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        feats = model(x)
    gouts = [g.cuda() for g in bc.full_gouts(name, [f.shape for f in feats])]
    sum((f.float() * g).sum() for f, g in zip(feats, gouts)).backward()
    res[amp] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None}
rows = []
for k, g in res[False].items():
    n = float(g.norm())
    if k in res[True] and n > 0:
        rows.append((float((res[True][k] - g).norm()) / n, k, n, float(res[True][k].norm())))
    elif k not in res[True]:
        rows.append((9.0, k, n, -1.0))
rows.sort(reverse=True)
for r in rows[:25]:
    print('%8.4f  %-60s |g32| %.4e  |g16| %.4e' % r)
