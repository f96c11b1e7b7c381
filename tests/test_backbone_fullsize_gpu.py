"""GPU: the headline workloads at their REAL sizes against digests made by the reference's own ViTAdapter classes
(tools/gen_golden_fullsize.py -> tests/golden/backbone_fullsize.npz):
BASELINE configs[1] (ViT-Adapter-T, 512 x 512, batch 2) and configs[2] (ViT-Adapter-B det flavour, 1024 x 1024, one
image), train mode, drop_path 0, seeded weights / input / output gradients (oracle/backbone_cases.py).

fp32 HIP path: every pyramid level within 1e-3 of the level's max at 4096 sampled positions and in its sum;
d(loss)/d(image) likewise but for the stem's max-pool arg-max flips; every parameter gradient's two digests and norm
within 8e-3 (digests) / 2e-3 (norm) of its scale (5e-2 below the max-pool).  bf16 autocast (the mode bench.py runs): levels within 3e-2
relative L2 of the reference samples; parameter gradients against the fp32 HIP run of the same process (itself pinned
by the digests): median relative L2 <= 8e-2, every one <= 0.25.
Also here: the configs[4] backbone (ViT-Adapter-L at 800 x 1344, one image) under bf16 autocast."""
import json
import os

os.environ.setdefault('MIOPEN_FIND_MODE', '2')      # the fp32 runs go through MIOpen convolutions: no exhaustive search per new shape

import numpy as np
import pytest
import torch

from oracle import backbone_cases as bc
from oracle import seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _fp32_math():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    yield


def _model(name):
    from vitadapter.backbones import ViTAdapter
    case = bc.FULLSIZE_CASES[name]
    m = ViTAdapter(**case['cfg'])
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(seeded.seeded_state_dict(shapes, 5))
    return m.cuda().train(), shapes


def _sampled(key, t):
    flat = t.detach().reshape(-1)
    pos = bc.fullsize_positions(key, flat.numel()).to(flat.device)
    return flat[pos].double().cpu().numpy()


def _run(model, name, amp, want_gx):
    model.zero_grad(set_to_none=True)
    x = bc.fullsize_input(name).cuda().requires_grad_(want_gx)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        outs = model(x)
    gouts = [g.cuda() for g in bc.fullsize_gouts(name, [o.shape for o in outs])]
    sum((o.float() * g).sum() for o, g in zip(outs, gouts)).backward()
    grads = {k: p.grad.detach().double() for k, p in model.named_parameters() if p.grad is not None}
    return [o.detach() for o in outs], (x.grad.detach() if want_gx else None), grads


@pytest.mark.parametrize('name', sorted(bc.FULLSIZE_CASES))
def test_fullsize_fp32_and_bf16_against_reference_digests(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, 'backbone_fullsize.npz'))
    case = bc.FULLSIZE_CASES[name]
    model, shapes = _model(name)
    meta = json.loads(str(gold['meta']))[name]
    assert {k: list(s) for k, s in shapes.items()} == meta['state_dict']
    H, W = case['hw']

    # ---- fp32
    outs, gx, g32 = _run(model, name, amp=False, want_gx=True)
    assert [tuple(o.shape) for o in outs] == [(case['batch'], case['cfg']['embed_dim'], H // s, W // s) for s in (4, 8, 16, 32)]
    for k, o in enumerate(outs):
        tag = '%s_f%d' % (name, k + 1)
        want = gold[tag + '_samples']
        s_sum, s_max, s_l2 = gold[tag + '_sum']
        got = _sampled(tag, o)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 1e-3 * max(1.0, s_max), (tag, np.abs(got - want).max(), s_max)
        assert abs(float(o.double().sum()) - s_sum) <= 1e-3 * s_l2, tag
        assert abs(float(o.double().pow(2).sum().sqrt()) - s_l2) <= 1e-3 * s_l2, tag
    # d(loss)/d(image): behind the stem's 3x3 max-pool, whose arg-max flips where two window entries tie to ~1e-6
    want = gold[name + '_gx_samples']
    got = _sampled(name + '_gx', gx)
    s_sum, s_max, s_l2 = gold[name + '_gx_sum']
    err = np.abs(got - want)
    assert float((err > 1e-3 * max(1.0, s_max)).mean()) <= 0.02
    assert np.sqrt((err ** 2).sum() / (want ** 2).sum()) <= 0.05
    checked = 0
    top = max(float(gold[key][2]) for key in gold.files if key.startswith(name + '_gp_'))
    for k, g in g32.items():
        key = '%s_gp_%s' % (name, k)
        assert key in gold.files, key
        d0, d1, norm = gold[key]
        if norm <= 1e-5 * top:          # exact gradient zero (a bias in front of a BatchNorm: up.bias, spm.fc1.bias): rounding noise
            continue
        # the digests are sums over up to 10^6 elements: an error of fp32 rounding size shows relative to the larger of the
        # gradient's norm and the sum itself (measured worst 4.4e-3, a mlp.fc2.weight whose elements share a sign)
        scale = max(norm, abs(d0), abs(d1))
        tol = 5e-2 if k.startswith('spm.stem') else 8e-3
        got_d = seeded.digest(g)
        assert abs(got_d[0] - d0) <= tol * scale and abs(got_d[1] - d1) <= tol * scale, (k, got_d, (d0, d1), norm)
        assert abs(float(g.norm()) - norm) <= (5e-2 if k.startswith('spm.stem') else 2e-3) * norm, (k, float(g.norm()), norm)
        checked += 1
    assert checked > (150 if name.startswith('tiny') else 300)

    # ---- bf16 autocast, the mode bench.py runs
    outs16, _, g16 = _run(model, name, amp=True, want_gx=False)
    for k, o in enumerate(outs16):
        tag = '%s_f%d' % (name, k + 1)
        want = gold[tag + '_samples']
        got = _sampled(tag, o.float())
        rel = np.sqrt(((got - want) ** 2).sum() / (want ** 2).sum())
        assert np.isfinite(got).all() and rel <= 3e-2, (tag, rel)
    top = max(float(g.norm()) for g in g32.values())
    rels = {}
    for k, g in g32.items():
        n = float(g.norm())
        # left out: the stem below the max-pool and parameters whose exact gradient is zero (a bias in front of a BatchNorm)
        if k.startswith('spm.stem') or n <= 1e-5 * top:
            continue
        rels[k] = float((g16[k] - g).norm()) / n
    vals = sorted(rels.values())
    assert len(vals) > 100 and vals[len(vals) // 2] <= 8e-2 and vals[-1] <= 0.25, (
        vals[len(vals) // 2], sorted(rels.items(), key=lambda kv: -kv[1])[:4])


def test_large_seg_800x1344_bf16(monkeypatch):
    """BASELINE configs[4] backbone: ViT-Adapter-L (embed 1024, depth 24, 16 heads, deform heads 16 x 32, ratio 0.5) at
    800 x 1344 (1333 x 800 padded to a multiple of 32), one image, train mode, bf16 autocast: non-square pyramid shapes,
    finite gradients, and the fused MSDeformAttn core + tile-pass backward against the reference's op sequence around
    the plain fp32 Function (VAH_MSDA_FUSED=0) on the same weights."""
    from vitadapter.backbones.vit_adapter import PRESETS, ViTAdapter
    kw = dict(PRESETS['large_seg'])
    kw['drop_path_rate'] = 0.0
    torch.manual_seed(0)
    model = ViTAdapter(**kw).cuda().train()
    x = torch.randn(1, 3, 800, 1344, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))
    feats = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('VAH_MSDA_FUSED', fused)
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            outs = model(x)
        assert [tuple(o.shape) for o in outs] == [(1, 1024, 200, 336), (1, 1024, 100, 168), (1, 1024, 50, 84), (1, 1024, 25, 42)]
        sum(o.float().pow(2).mean() for o in outs).backward()
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        assert len(grads) > 500 and all(torch.isfinite(g).all() for g in grads)
        feats[fused] = [o.detach().float() for o in outs]
    for a, b in zip(feats['1'], feats['0']):
        assert float((a - b).abs().max()) <= 3e-2 * max(1.0, float(b.abs().max()))
