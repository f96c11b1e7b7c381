"""Goldens for vitadapter/checkpoint.py from the reference's own load_checkpoint.  Container only.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_checkpoint.py

Imports /root/reference/segmentation/mmcv_custom/checkpoint.py (file loaded directly; stand-ins for the absent mmcv /
torchvision modules it names at import time: FileClient, load, is_module_wrapper -> False, get_dist_info -> (0, 1),
mkdir_or_exist) and the reference's BEiT class (base/beit.py, under the stand-ins of tools/gen_golden.py), writes
seeded synthetic checkpoints (oracle/checkpoint_cases.py) to a temporary directory with torch.save, lets the
reference load them into its model and stores the resulting tensors of the keys the loader rewrites in
tests/golden/checkpoint.npz.

Cases the reference can run here: pos_embed bicubic resize (:457-484), shared relative-position-bias expansion
(:375-388), `relative_position_index` drop (:391-393), `module.` prefix / `state_dict` wrapper (:341-355), same-size
tables.  NOT generated: the geometric table resize (:395-455) - it calls scipy.interpolate.interp2d, which this
image's SciPy (1.15) no longer has; that branch stays "parity unpinned" (tests/test_checkpoint.py holds it to
properties).
"""
import importlib
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import gen_golden as gg                                   # noqa: E402
from oracle import checkpoint_cases as cc                 # noqa: E402


def load_reference_loader_det():
    path = os.path.join(gg.REF, 'detection', 'mmcv_custom', 'checkpoint.py')
    spec = importlib.util.spec_from_file_location('ref_checkpoint_det', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_loader():
    gg._mod('mmcv.fileio', FileClient=object, load=lambda *a, **k: None)
    gg._mod('mmcv.parallel', is_module_wrapper=lambda m: False)
    sys.modules['mmcv.runner'].get_dist_info = lambda: (0, 1)
    gg._mod('mmcv.utils', mkdir_or_exist=lambda *a, **k: None)
    gg._mod('torchvision')
    path = os.path.join(gg.REF, 'segmentation', 'mmcv_custom', 'checkpoint.py')
    spec = importlib.util.spec_from_file_location('ref_checkpoint', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    gg.load_reference_ops()
    layers = sys.modules['timm.models.layers']
    layers.drop_path = lambda x, drop_prob=0., training=False: x
    sys.modules['mmcv_custom'].load_checkpoint = lambda *a, **k: None
    gg.load_reference_backbone('seg')
    beit = importlib.import_module('ref_seg.base.beit')
    ref = load_reference_loader()
    g = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, case in cc.CASES.items():
            torch.manual_seed(0)
            model = beit.BEiT(**case['model'])
            for p in model.parameters():                  # known start values: untouched keys must stay as they are
                torch.nn.init.constant_(p, 0.25)
            path = os.path.join(tmp, name + '.pth')
            torch.save(cc.checkpoint(name), path)
            ref.load_checkpoint(model, path, map_location='cpu', strict=False, logger=None)
            sd = model.state_dict()
            for k in case['check']:
                g['%s/%s' % (name, k)] = sd[k].detach().numpy()
        # detection flavour: the reference's det loader into the reference's det BEiT (windowed / global blocks)
        gg.load_reference_backbone('det')
        beit_det = importlib.import_module('ref_det.base.beit')
        ref_det = load_reference_loader_det()
        for name, case in cc.DET_CASES.items():
            torch.manual_seed(0)
            model = beit_det.BEiT(**case['model'])
            for p in model.parameters():
                torch.nn.init.constant_(p, 0.25)
            path = os.path.join(tmp, name + '.pth')
            torch.save(cc.checkpoint(name), path)
            ref_det.load_checkpoint(model, path, map_location='cpu', strict=False, logger=None)
            sd = model.state_dict()
            for k in case['check']:
                g['%s/%s' % (name, k)] = sd[k].detach().numpy()
    dst = os.path.join(ROOT, 'tests', 'golden', 'checkpoint.npz')
    np.savez_compressed(dst, **g)
    print(dst, os.path.getsize(dst), 'bytes,', len(g), 'arrays')


if __name__ == '__main__':
    main()
