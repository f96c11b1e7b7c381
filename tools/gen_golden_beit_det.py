"""BEiTAdapter goldens from the reference's own class, DETECTION flavour.  Container only.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_beit_det.py

Imports /root/reference/detection/mmdet_custom/models/backbones/{base/beit.py, adapter_modules.py, beit_adapter.py}
under the in-memory stand-ins of tools/gen_golden.py (timm / mmcv / mmdet; the MSDA extension call replaced by the
reference's own pure-PyTorch core), feeds seeded weights and inputs (oracle/seeded.py,
oracle/backbone_cases.py::BEIT_DET_CASES) and stores the expected OUTPUTS in tests/golden/beit_adapter_det.npz: the four
maps, the input gradient, digests of every parameter gradient, and the state_dict key -> shape table.
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import gen_golden as gg                                   # noqa: E402
from oracle import backbone_cases as bc                   # noqa: E402
from oracle import seeded                                 # noqa: E402


def main():
    gg.load_reference_ops()
    layers = sys.modules['timm.models.layers']

    def drop_path(x, drop_prob=0., training=False):      # timm 0.4.12 functional form
        if drop_prob == 0. or not training:
            return x
        keep = 1 - drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x.div(keep) * mask
    layers.drop_path = drop_path
    sys.modules['mmcv_custom'].load_checkpoint = lambda *a, **k: None
    gg.load_reference_backbone('det')
    ba = importlib.import_module('ref_det.beit_adapter')
    g, meta = {}, {}
    for name, case in bc.BEIT_DET_CASES.items():
        torch.manual_seed(0)
        model = ba.BEiTAdapter(**case['cfg'])
        meta[name] = {k: list(v.shape) for k, v in model.state_dict().items()}
        missing, unexpected = model.load_state_dict(seeded.seeded_state_dict(bc.float_shapes(model), 23), strict=False)
        assert not unexpected and all(k.endswith('relative_position_index') for k in missing), (missing, unexpected)
        for mode in case['modes']:
            model.train(mode == 'train')
            model.zero_grad(set_to_none=True)
            x = bc.beit_det_input(name).requires_grad_(True)
            outs = model(x)
            gouts = bc.beit_det_gouts(name, [o.shape for o in outs])
            params = list(model.parameters())
            grads = torch.autograd.grad(sum((o * go).sum() for o, go in zip(outs, gouts)), [x] + params, allow_unused=True)
            tag = '%s_%s' % (name, mode)
            for k, o in enumerate(outs):
                g['%s_f%d' % (tag, k + 1)] = o.detach().numpy()
            g[tag + '_gx'] = grads[0].numpy()
            for (pn, _), gr in zip(model.named_parameters(), grads[1:]):
                if gr is not None:
                    g['%s_gp_%s' % (tag, pn)] = seeded.digest(gr)
    g['meta'] = np.array(json.dumps(meta))
    dst = os.path.join(ROOT, 'tests', 'golden', 'beit_adapter_det.npz')
    np.savez_compressed(dst, **g)
    print(dst, os.path.getsize(dst), 'bytes,', len(g), 'arrays')


if __name__ == '__main__':
    main()
