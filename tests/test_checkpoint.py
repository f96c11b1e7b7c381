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
