#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE's own Python (container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [msda] [module] [backbone]

The reference tree (/root/reference) is imported in-process with in-memory stand-ins for
the third-party packages that are absent here (timm, mmcv, mmseg/mmdet and the compiled
MultiScaleDeformableAttention extension); SURVEY.md appendix A is the recipe.  The
reference's MSDeformAttnFunction.apply is pointed at the reference's own
ms_deform_attn_core_pytorch so autograd yields reference gradients on CPU.

Only seeded INPUT builders from oracle/cases.py + oracle/seeded.py and the reference's
outputs are involved; the fixtures hold expected outputs (+ digests of the regenerated
inputs), never reference source.  This script never runs on the GPU box.
"""
import importlib
import json
import os
import sys
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
from oracle import cases, seeded  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


# --------------------------------------------------------------------------------------
# stand-ins for absent third-party modules
# --------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):           # timm 0.4.12 DropPath: identity when p == 0 or eval
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x.div(keep) * mask


class _Mlp(nn.Module):                # timm 0.4.12 Mlp: fc1 -> act -> drop -> fc2 -> drop
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def install_stubs():
    _mod('MultiScaleDeformableAttention')
    _mod('timm'), _mod('timm.models')
    _mod('timm.models.layers', DropPath=_DropPath, Mlp=_Mlp,
         to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
         trunc_normal_=torch.nn.init.trunc_normal_)
    _mod('mmcv'), _mod('mmcv.runner', BaseModule=nn.Module)
    for pkg in ('mmseg', 'mmdet'):
        _mod(pkg), _mod(pkg + '.models')
        _mod(pkg + '.models.builder', BACKBONES=_Registry())
        _mod(pkg + '.utils', get_root_logger=lambda *a, **k: None)
    _mod('mmcv_custom', my_load_checkpoint=lambda *a, **k: None)
    sys.path.insert(0, os.path.join(REF, 'detection'))        # -> import ops.*


def load_reference_ops():
    install_stubs()
    func = importlib.import_module('ops.functions.ms_deform_attn_func')
    mods = importlib.import_module('ops.modules.ms_deform_attn')

    class _CpuFunction:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return func.ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    mods.MSDeformAttnFunction = _CpuFunction
    return func, mods


def load_reference_backbone(flavour):
    """flavour: 'seg' | 'det' -> (vit module, adapter_modules module, vit_adapter module)."""
    sub = {'seg': 'segmentation/mmseg_custom', 'det': 'detection/mmdet_custom'}[flavour]
    d = os.path.join(REF, sub, 'models', 'backbones')
    pkg = 'ref_' + flavour
    p = _mod(pkg)
    p.__path__ = [d]
    b = _mod(pkg + '.base')
    b.__path__ = [os.path.join(d, 'base')]
    vit = importlib.import_module(pkg + '.base.vit')
    am = importlib.import_module(pkg + '.adapter_modules')
    va = importlib.import_module(pkg + '.vit_adapter')
    return vit, am, va


def _np(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------------------
# G1 + G2: the op itself
# --------------------------------------------------------------------------------------
def gen_msda():
    func, _ = load_reference_ops()
    core = func.ms_deform_attn_core_pytorch

    def run(value, hw, loc, attn, gout):
        value = value.clone().requires_grad_(True)
        loc = loc.clone().requires_grad_(True)
        attn = attn.clone().requires_grad_(True)
        out = core(value, hw, loc, attn)
        out.backward(gout)
        return out, value.grad, loc.grad, attn.grad

    # G1: detection/ops/test.py shapes, fp64, every channel count it lists
    g1 = {}
    for D in cases.TESTPY_CHANNELS:
        value, hw, lsi, loc, attn, gout = cases.testpy_inputs(D)
        out, gv, gl, ga = run(value, hw, loc, attn, gout)
        g1['D%d_out' % D] = _np(out)
        g1['D%d_gv' % D] = _np(gv)
        g1['D%d_gl' % D] = _np(gl)
        g1['D%d_ga' % D] = _np(ga)
        g1['D%d_digest' % D] = np.concatenate([seeded.digest(t) for t in (value, loc, attn, gout)])
    np.savez_compressed(os.path.join(GOLD, 'msda_testpy.npz'), **g1)

    # G2: adapter-shaped calls; fp32 inputs, reference evaluated in fp64 (truth) and fp32
    g2 = {}
    for name, kw in cases.ADAPTER_CASES.items():
        value, hw, lsi, loc, attn, gout = cases.msda_inputs(name, **kw)
        out, gv, gl, ga = run(value.double(), hw, loc.double(), attn.double(), gout.double())
        out32, gv32, gl32, ga32 = run(value, hw, loc, attn, gout)
        # stored as fp32 roundings of the fp64 results (keeps the fixture small; the tight
        # fp64 pin of the oracle is G1 above)
        g2[name + '_out'] = _np(out).astype(np.float32)
        g2[name + '_gv'] = _np(gv).astype(np.float32)
        g2[name + '_gl'] = _np(gl).astype(np.float32)
        g2[name + '_ga'] = _np(ga).astype(np.float32)
        # how far the reference's own fp32 evaluation sits from its fp64 one (context for tolerances)
        g2[name + '_ref32_err'] = np.array([
            (out32.double() - out).abs().max().item(), (gv32.double() - gv).abs().max().item(),
            (gl32.double() - gl).abs().max().item(), (ga32.double() - ga).abs().max().item()])
        g2[name + '_digest'] = np.concatenate([seeded.digest(t) for t in (value, loc, attn, gout)])
    np.savez_compressed(os.path.join(GOLD, 'msda_adapter.npz'), **g2)
    print('msda goldens written')


# --------------------------------------------------------------------------------------
# G3: the MSDeformAttn module
# --------------------------------------------------------------------------------------
MODULE_CASES = {
    # name: (d_model, n_levels, n_heads, n_points, ratio, Lq, value shapes, query shapes, N)
    'inj_t': (192, 3, 6, 4, 1.0, 16, [(8, 8), (4, 4), (2, 2)], [(4, 4)], 2),
    'ext_t': (192, 1, 6, 4, 1.0, 84, [(4, 4)], [(8, 8), (4, 4), (2, 2)], 2),
    'inj_b': (768, 3, 12, 4, 0.5, 16, [(8, 8), (4, 4), (2, 2)], [(4, 4)], 1),
    'ext_b': (768, 1, 12, 4, 0.5, 84, [(4, 4)], [(8, 8), (4, 4), (2, 2)], 1),
}


def module_inputs(name):
    d_model, L, M, P, ratio, Lq, vshapes, qshapes, N = MODULE_CASES[name]
    S = sum(h * w for h, w in vshapes)
    query = seeded.randn('module/%s/query' % name, (N, Lq, d_model), 1)
    feat = seeded.randn('module/%s/feat' % name, (N, S, d_model), 1)
    ref = cases.reference_grid(qshapes)
    hw = torch.as_tensor(vshapes, dtype=torch.long)
    gout = seeded.randn('module/%s/gout' % name, (N, Lq, d_model), 1)
    return query, ref, feat, hw, cases.level_start_index(vshapes), gout


def gen_module():
    _, mods = load_reference_ops()
    g = {}
    meta = {}
    for name, (d_model, L, M, P, ratio, Lq, vshapes, qshapes, N) in MODULE_CASES.items():
        m = mods.MSDeformAttn(d_model=d_model, n_levels=L, n_heads=M, n_points=P, ratio=ratio)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(seeded.seeded_state_dict(shapes, seed=2))
        query, ref, feat, hw, lsi, gout = module_inputs(name)
        query.requires_grad_(True)
        feat.requires_grad_(True)
        out = m(query, ref, feat, hw, lsi, None)
        out.backward(gout)
        g[name + '_out'] = _np(out)
        g[name + '_gquery'] = _np(query.grad)
        g[name + '_gfeat'] = _np(feat.grad)
        for k, p in m.named_parameters():
            g['%s_gparam_%s' % (name, k)] = seeded.digest(p.grad)
        meta[name] = {k: list(s) for k, s in shapes.items()}
    g['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, 'msda_module.npz'), **g)
    print('module goldens written')


if __name__ == '__main__':
    what = sys.argv[1:] or ['msda', 'module']
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if 'msda' in what:
        gen_msda()
    if 'module' in what:
        gen_module()
    if 'backbone' in what:
        from gen_golden_backbone import gen_backbone
        gen_backbone()
