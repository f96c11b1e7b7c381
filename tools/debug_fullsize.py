import json, os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', '2')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import backbone_cases as bc, seeded
import test_backbone_fullsize_gpu as T
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'backbone_fullsize.npz'))
for name in sys.argv[1:]:
    t0 = time.time()
    model, shapes = T._model(name)
    print(name, 'model %.1fs' % (time.time() - t0), flush=True)
    outs, gx, g32 = T._run(model, name, False, True)
    torch.cuda.synchronize(); print('fp32 run %.1fs' % (time.time() - t0), flush=True)
    for k, o in enumerate(outs):
        tag = '%s_f%d' % (name, k + 1)
        got = T._sampled(tag, o); want = gold[tag + '_samples']; s = gold[tag + '_sum']
        print(tag, 'max err/max %.2e' % (np.abs(got - want).max() / max(1, s[1])), 'sum err/l2 %.2e' % (abs(float(o.double().sum()) - s[0]) / s[2]))
    worst = []
    for k, g in g32.items():
        d0, d1, norm = gold['%s_gp_%s' % (name, k)]
        gd = seeded.digest(g)
        worst.append((max(abs(gd[0] - d0), abs(gd[1] - d1)) / max(norm, abs(d0), abs(d1)), abs(float(g.norm()) - norm) / norm, k))
    worst.sort(reverse=True)
    print('digest worst:', worst[:6]); print('norm worst:', sorted(worst, key=lambda t: -t[1])[:4], flush=True)
    outs16, _, g16 = T._run(model, name, True, False)
    torch.cuda.synchronize(); print('bf16 run %.1fs' % (time.time() - t0), flush=True)
    for k, o in enumerate(outs16):
        tag = '%s_f%d' % (name, k + 1)
        got = T._sampled(tag, o.float()); want = gold[tag + '_samples']
        print(tag, 'bf16 rel l2 %.3e' % np.sqrt(((got - want) ** 2).sum() / (want ** 2).sum()))
    top = max(float(g.norm()) for g in g32.values())
    rels = sorted(((float((g16[k] - g).norm()) / float(g.norm()), k) for k, g in g32.items() if not k.startswith('spm.stem') and float(g.norm()) > 1e-5 * top), reverse=True)
    print('bf16 grads: median %.3f' % rels[len(rels) // 2][0], 'worst', rels[:6], flush=True)
