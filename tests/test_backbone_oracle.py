"""CPU: (1) the product ViTAdapter's state_dict layout equals the reference's key for key;
(2) the oracle restatement of the backbone (oracle/vit_adapter_ref.py) reproduces the outputs
and gradients that the reference's own classes produced (tests/golden/backbone.npz, made by
tools/gen_golden_backbone.py).  The HIP-backed product is then checked against this oracle on
the GPU (tests/test_backbone_gpu.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbone_cases as bc
from oracle import seeded
from oracle import vit_adapter_ref as ref


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'backbone.npz'))


def _close(got, want, tol, what):
    got = got.detach().double().numpy() if torch.is_tensor(got) else got
    want = want.astype(np.float64)
    assert got.shape == want.shape, what
    err = np.abs(got - want).max()
    assert err <= tol * max(1.0, np.abs(want).max()), '%s: %.3e' % (what, err)


@pytest.mark.parametrize('name', sorted(bc.FULL_CASES))
def test_state_dict_layout_matches_reference(gold, name):
    from vitadapter.backbones import ViTAdapter
    meta = json.loads(str(gold['meta']))[name]
    model = ViTAdapter(**bc.FULL_CASES[name]['cfg'])
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(mine.keys()) == list(meta.keys())        # same keys, same order
    assert mine == meta


@pytest.mark.parametrize('name', sorted(bc.FULL_CASES))
def test_oracle_backbone_matches_reference(gold, name):
    case = bc.FULL_CASES[name]
    meta = json.loads(str(gold['meta']))[name]
    cfg = ref.Cfg(**case['cfg'])
    for mode in case['modes']:
        sd = seeded.seeded_state_dict({k: tuple(s) for k, s in meta.items()}, 5)
        for k, v in sd.items():
            if v.is_floating_point() and 'running_' not in k:
                v.requires_grad_(True)
        x = bc.full_input(name).requires_grad_(True)
        outs = ref.vit_adapter_forward(sd, x, cfg, training=(mode == 'train'))
        H, W = case['hw']
        assert [tuple(o.shape[-2:]) for o in outs] == [(H // 4, W // 4), (H // 8, W // 8),
                                                       (H // 16, W // 16), (H // 32, W // 32)]
        tag = '%s_%s' % (name, mode)
        for k, o in enumerate(outs):
            _close(o, gold['%s_f%d' % (tag, k + 1)], 2e-5, '%s f%d' % (tag, k + 1))
        gouts = bc.full_gouts(name, [o.shape for o in outs])
        loss = sum((o * g).sum() for o, g in zip(outs, gouts))
        names = [k for k, v in sd.items() if v.requires_grad]
        grads = torch.autograd.grad(loss, [x] + [sd[k] for k in names], allow_unused=True)
        _close(grads[0], gold[tag + '_gx'], 5e-5, tag + ' grad x')
        checked = 0
        for k, gr in zip(names, grads[1:]):
            key = '%s_gp_%s' % (tag, k)
            if key in gold.files and gr is not None:
                want = gold[key]
                got = seeded.digest(gr)
                assert np.abs(got - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), key
                checked += 1
        assert checked > 100


def _part_sd(module, seed):
    return seeded.seeded_state_dict({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed)


def test_oracle_parts_match_reference(gold):
    from vitadapter.backbones import adapter_modules as am
    from vitadapter.backbones import vit
    E, M, R = bc.PART['embed'], bc.PART['deform_heads'], bc.PART['ratio']
    H = W = bc.PART['tokens']
    geo1, geo2 = bc.part_geometry()
    cfg = ref.Cfg(embed_dim=E, deform_num_heads=M, deform_ratio=R, n_points=4, num_heads=bc.PART['heads'])
    x, c = bc.part_tokens()

    sd = _part_sd(am.Injector(dim=E, n_levels=3, num_heads=M, n_points=4, deform_ratio=R), 6)
    xi, ci = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
    o = ref.injector(sd, xi, ci, geo1[0], [tuple(s) for s in geo1[1].tolist()], cfg)
    g = torch.autograd.grad((o * bc.part_gout('inj', o.shape)).sum(), [xi, ci])
    _close(o, gold['part_inj_out'], 1e-5, 'inj out')
    _close(g[0], gold['part_inj_gx'], 2e-5, 'inj gx')
    _close(g[1], gold['part_inj_gc'], 2e-5, 'inj gc')

    sd = _part_sd(am.Extractor(dim=E, num_heads=M, n_points=4, n_levels=1, deform_ratio=R), 7)
    xi, ci = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
    o = ref.extractor(sd, ci, xi, geo2[0], [tuple(s) for s in geo2[1].tolist()], H, W, cfg)
    g = torch.autograd.grad((o * bc.part_gout('ext', o.shape)).sum(), [xi, ci])
    _close(o, gold['part_ext_out'], 1e-5, 'ext out')
    _close(g[0], gold['part_ext_gx'], 2e-5, 'ext gx')
    _close(g[1], gold['part_ext_gc'], 2e-5, 'ext gc')

    sd = _part_sd(am.SpatialPriorModule(inplanes=bc.PART['inplanes'], embed_dim=E), 8)
    for mode in ('eval', 'train'):
        img = bc.part_image().requires_grad_(True)
        outs = ref.spm(sd, img, mode == 'train')
        g = torch.autograd.grad(sum((o * bc.part_gout('spm%d' % k, o.shape)).sum()
                                    for k, o in enumerate(outs)), [img])
        for k, o in enumerate(outs):
            _close(o, gold['part_spm_%s_c%d' % (mode, k + 1)], 1e-5, 'spm c%d' % (k + 1))
        _close(g[0], gold['part_spm_%s_gimg' % mode], 5e-5, 'spm gimg')

    for bname, (windowed, Hb, Wb) in bc.BLOCK_CASES.items():
        sd = _part_sd(vit.Block(dim=E, num_heads=bc.PART['heads'], mlp_ratio=4., qkv_bias=True,
                                windowed=windowed, window_size=14, layer_scale=True), 9)
        t = bc.block_tokens(bname).requires_grad_(True)
        o = ref.block(sd, t, Hb, Wb, bc.PART['heads'], windowed, 14, True)
        g = torch.autograd.grad((o * bc.part_gout(bname, o.shape)).sum(), [t])
        _close(o, gold['part_%s_out' % bname], 1e-5, bname)
        _close(g[0], gold['part_%s_gx' % bname], 2e-5, bname + ' gx')


def test_oracle_backbone_matches_reference_at_full_size(golden_dir):
    """BASELINE configs[1] at its real size (ViT-Adapter-T, 512 x 512, batch 2, train mode): the oracle restatement
    against the digests the reference's own class produced (tools/gen_golden_fullsize.py) - 4096 sampled elements, sum
    and L2 norm of every pyramid level.  (configs[2] at 1024 x 1024 takes a minute on the CPU: GPU tier only.)"""
    gold = np.load(os.path.join(golden_dir, 'backbone_fullsize.npz'))
    name = 'tiny_seg_512'
    case = bc.FULLSIZE_CASES[name]
    meta = json.loads(str(gold['meta']))[name]['state_dict']
    sd = seeded.seeded_state_dict({k: tuple(s) for k, s in meta.items()}, 5)
    with torch.no_grad():
        outs = ref.vit_adapter_forward(sd, bc.fullsize_input(name), ref.Cfg(**case['cfg']), training=True)
    for k, o in enumerate(outs):
        tag = '%s_f%d' % (name, k + 1)
        flat = o.reshape(-1)
        got = flat[bc.fullsize_positions(tag, flat.numel())].double().numpy()
        s_sum, s_max, s_l2 = gold[tag + '_sum']
        assert np.abs(got - gold[tag + '_samples']).max() <= 2e-4 * max(1.0, s_max), tag
        assert abs(float(o.double().sum()) - s_sum) <= 2e-4 * s_l2 and abs(float(o.double().pow(2).sum().sqrt()) - s_l2) <= 2e-4 * s_l2, tag
