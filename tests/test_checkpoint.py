"""vitadapter/checkpoint.py against the reference's own load_checkpoint
(/root/reference/segmentation/mmcv_custom/checkpoint.py:319-516): the goldens in tests/golden/checkpoint.npz are the
tensors the REFERENCE loader left in the reference's BEiT model for the seeded checkpoints of
oracle/checkpoint_cases.py (tools/gen_golden_checkpoint.py).  CPU only.

The geometric relative-position-bias resize (:395-455) is PARITY UNPINNED: the reference calls
scipy.interpolate.interp2d, which this image's SciPy no longer has, so no golden can be made; it is held to
properties of the documented algorithm instead (identity at equal sizes, class-token rows kept, exact reproduction of
polynomials of the geometric coordinates - an interpolating cubic spline reproduces cubics)."""
import os

import numpy as np
import torch

from oracle import checkpoint_cases as cc
from vitadapter import checkpoint as ck
from vitadapter.backbones.beit import BEiT


def _model(case):
    torch.manual_seed(0)
    m = BEiT(**case['model'])
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(0.25)
    return m


def test_loader_matches_reference_goldens(golden_dir, tmp_path):
    gold = np.load(os.path.join(golden_dir, 'checkpoint.npz'))
    for name, case in cc.CASES.items():
        model = _model(case)
        path = str(tmp_path / (name + '.pth'))
        torch.save(cc.checkpoint(name), path)
        ck.load_checkpoint(model, path)
        sd = model.state_dict()
        for k in case['check']:
            want = gold['%s/%s' % (name, k)]
            got = sd[k].detach().numpy()
            assert got.shape == want.shape, (name, k)
            assert np.abs(got.astype(np.float64) - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (name, k)


def test_strict_reports_missing_keys(tmp_path):
    case = cc.CASES['pos_embed_resize']
    path = str(tmp_path / 'c.pth')
    torch.save(cc.checkpoint('pos_embed_resize'), path)
    try:
        ck.load_checkpoint(_model(case), path, strict=True)
    except RuntimeError as e:
        assert 'missing keys in source state_dict' in str(e) and 'cls_token' in str(e)
    else:
        raise AssertionError('strict load of a partial checkpoint must raise')


def test_geometric_rel_pos_bias_resize_properties():
    heads, extra = 3, 3
    src, dst = 7, 9                                              # 4x4 -> 5x5 patches: (2p - 1)^2 table entries
    x, dx = ck._geometric_coordinates(src, dst)
    assert len(x) == src and len(dx) == dst and x[src // 2] == 0.0 and abs(x[-1] - dst // 2) < 1e-3
    gy, gx = np.meshgrid(x, x, indexing='ij')
    coeff = [(0.3, -1.2, 0.7, 0.05, -0.02), (1.0, 0.0, 0.0, 0.0, 0.0), (-0.5, 0.4, 0.1, -0.03, 0.01)]
    body = np.stack([a + b * gx + c * gy + d * gx * gy + e * gx ** 3 for a, b, c, d, e in coeff], -1).reshape(src * src, heads)
    tail = np.arange(extra * heads, dtype=np.float64).reshape(extra, heads)
    table = torch.tensor(np.concatenate([body, tail]), dtype=torch.float32)
    out = ck.resize_rel_pos_bias_table(table, dst * dst + extra, (5, 5))
    assert out.shape == (dst * dst + extra, heads)
    assert torch.equal(out[-extra:], table[-extra:])                             # class-token rows untouched
    dy, dxx = np.meshgrid(dx, dx, indexing='ij')
    want = np.stack([a + b * dxx + c * dy + d * dxx * dy + e * dxx ** 3 for a, b, c, d, e in coeff], -1).reshape(dst * dst, heads)
    assert np.abs(out[:-extra].numpy() - want).max() <= 1e-4 * np.abs(want).max()
    same = ck.resize_rel_pos_bias_table(table, src * src + extra, (4, 4))         # equal sizes: returned as is
    assert same is table


def _det_model(kw):
    from vitadapter.backbones.beit_det import BEiT as BEiTDet
    torch.manual_seed(0)
    m = BEiTDet(**kw)
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(0.25)
    return m


def test_det_loader_matches_reference_goldens(golden_dir, tmp_path):
    """Detection flavour (/root/reference/detection/mmcv_custom/checkpoint.py:379-445) against what the reference's det
    loader left in the reference's det BEiT: per-block tables, the checkpoint's 3 class-token rows always dropped."""
    import warnings
    gold = np.load(os.path.join(golden_dir, 'checkpoint.npz'))
    for name, case in cc.DET_CASES.items():
        model = _det_model(case['model'])
        path = str(tmp_path / (name + '.pth'))
        torch.save(cc.checkpoint(name), path)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ck.load_checkpoint(model, path, flavour='det')
        sd = model.state_dict()
        for k in case['check']:
            want = gold['%s/%s' % (name, k)]
            got = sd[k].detach().numpy()
            assert got.shape == want.shape, (name, k)
            assert np.abs(got.astype(np.float64) - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (name, k)


def test_det_loader_mixed_windows_from_a_27x27_table(tmp_path):
    """A real BEiT checkpoint has (27^2 + 3)-row tables (224 px, 14 x 14 patches); the detection model mixes windows of
    14 (27 x 27 rows: equal size, class rows dropped) and 56 (111 x 111 rows: geometric resize).  The seg formula
    (num_extra = dst - (2 ps - 1)^2) is wrong here: shapes, finiteness, and the equal-size rows bit for bit.
    The geometric branch itself is parity unpinned (module docstring)."""
    import warnings
    heads = 2
    kw = dict(img_size=224, patch_size=16, embed_dim=32, depth=2, num_heads=heads, mlp_ratio=2, qkv_bias=True,
              init_values=1e-6, drop_path_rate=0., use_abs_pos_emb=False, use_rel_pos_bias=True,
              window_attn=[True, False], window_size=[14, 56])
    model = _det_model(kw)
    own = model.state_dict()
    assert own['blocks.0.attn.relative_position_bias_table'].shape == (27 * 27, heads)
    assert own['blocks.1.attn.relative_position_bias_table'].shape == (111 * 111, heads)
    g = torch.Generator().manual_seed(5)
    table = torch.randn(27 * 27 + 3, heads, generator=g)
    path = str(tmp_path / 'beit.pth')
    torch.save({'model': {'rel_pos_bias.relative_position_bias_table': table}}, path)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ck.load_checkpoint(model, path, flavour='det')
    sd = model.state_dict()
    assert torch.equal(sd['blocks.0.attn.relative_position_bias_table'], table[:-3])
    big = sd['blocks.1.attn.relative_position_bias_table']
    assert big.shape == (111 * 111, heads) and torch.isfinite(big).all()
    # the spline interpolates: at the geometric source coordinates that are integers (0 and +-1) it returns the source
    centre = big.view(111, 111, heads)[55, 55]
    assert torch.allclose(centre, table[:-3].view(27, 27, heads)[13, 13], atol=1e-5)


def test_loader_warns_without_a_logger_and_skips_size_mismatches(tmp_path):
    """Reference behaviour (checkpoint.py:43-108): missing / unexpected keys are PRINTED when no logger is given, and a
    tensor of the wrong size is recorded and skipped instead of aborting the load.  A detector checkpoint's
    `backbone.` keys are taken, its neck / head keys dropped."""
    import warnings
    case = cc.CASES['pos_embed_resize']
    model = _model(case)
    C = case['model']['embed_dim']
    sd = {'backbone.blocks.0.attn.qkv.weight': torch.full((3 * C, C), 0.5), 'backbone.blocks.1.mlp.fc2.bias': torch.zeros(C + 1),
          'neck.lateral.weight': torch.zeros(4), 'decode_head.conv.weight': torch.zeros(2)}
    path = str(tmp_path / 'det.pth')
    torch.save({'state_dict': sd}, path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ck.load_checkpoint(model, path)
    text = ' '.join(str(x.message) for x in w)
    assert 'missing keys in source state_dict' in text and 'size mismatch (not loaded): blocks.1.mlp.fc2.bias' in text
    assert 'neck.' not in text
    got = model.state_dict()
    assert float(got['blocks.0.attn.qkv.weight'].mean()) == 0.5 and float(got['blocks.1.mlp.fc2.bias'].mean()) == 0.25
