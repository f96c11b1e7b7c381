#!/usr/bin/env python
"""Full-size goldens of the headline workloads from the REFERENCE's own ViTAdapter classes (container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_fullsize.py [tiny_seg_512] [base_det_1024]

BASELINE configs[1] (ViT-Adapter-T 512 x 512 batch 2) and configs[2] (ViT-Adapter-B det, 1024 x 1024, one image), train
mode, drop_path 0, seeded weights (oracle/seeded.py) and input (oracle/backbone_cases.py::FULLSIZE_CASES); the
reference classes are imported as tools/gen_golden.py does, MSDeformAttnFunction.apply pointed at
the reference's own ms_deform_attn_core_pytorch, everything fp32 on the CPU.  The fixture (tests/golden/
backbone_fullsize.npz) holds digests only: per output the fp64 sum, max |.| and 4096 sampled elements; the same for
d(loss)/d(image); per parameter gradient seeded.digest (2 fp64 numbers) and its L2 norm.  Never runs on the GPU box.
"""
import json
import os
import sys
import time

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch

from gen_golden import GOLD, load_reference_backbone, load_reference_ops
from oracle import backbone_cases as bc
from oracle import seeded


def _sampled(key, t):
    flat = t.detach().reshape(-1)
    pos = bc.fullsize_positions(key, flat.numel())
    return flat[pos].to(torch.float64).numpy()


def gen(names):
    load_reference_ops()
    path = os.path.join(GOLD, 'backbone_fullsize.npz')
    g = dict(np.load(path)) if os.path.exists(path) else {}
    meta = json.loads(str(g['meta'])) if 'meta' in g else {}
    for name in names:
        case = bc.FULLSIZE_CASES[name]
        _, _, va = load_reference_backbone(case['cfg']['flavour'])
        kw = {k: v for k, v in case['cfg'].items() if k != 'flavour'}
        t0 = time.time()
        model = va.ViTAdapter(**kw)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(seeded.seeded_state_dict(shapes, 5))
        model.train()
        x = bc.fullsize_input(name).requires_grad_(True)
        outs = model(x)
        print(name, 'forward %.0f s' % (time.time() - t0), [tuple(o.shape) for o in outs], flush=True)
        gouts = bc.fullsize_gouts(name, [o.shape for o in outs])
        loss = sum((o * go).sum() for o, go in zip(outs, gouts))
        params = list(model.named_parameters())
        grads = torch.autograd.grad(loss, [x] + [p for _, p in params], allow_unused=True)
        print(name, 'backward done %.0f s' % (time.time() - t0), flush=True)
        for k, o in enumerate(outs):
            tag = '%s_f%d' % (name, k + 1)
            g[tag + '_sum'] = np.array([o.detach().double().sum().item(), o.detach().abs().max().item(),
                                        o.detach().double().pow(2).sum().sqrt().item()])
            g[tag + '_samples'] = _sampled(tag, o)
        g[name + '_gx_sum'] = np.array([grads[0].double().sum().item(), grads[0].abs().max().item(),
                                        grads[0].double().pow(2).sum().sqrt().item()])
        g[name + '_gx_samples'] = _sampled(name + '_gx', grads[0])
        for (pn, _), gr in zip(params, grads[1:]):
            if gr is not None:
                g['%s_gp_%s' % (name, pn)] = np.concatenate([seeded.digest(gr), [gr.double().pow(2).sum().sqrt().item()]])
        meta[name] = {'state_dict': {k: list(s) for k, s in shapes.items()},
                      'torch': torch.__version__, 'seconds': round(time.time() - t0)}
        g['meta'] = np.array(json.dumps(meta))
        np.savez_compressed(path, **g)
        print(name, 'written', flush=True)


if __name__ == '__main__':
    gen(sys.argv[1:] or list(bc.FULLSIZE_CASES))
