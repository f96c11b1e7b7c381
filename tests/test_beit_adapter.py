"""BEiTAdapter against goldens produced by the reference's own BEiTAdapter class
(tools/gen_golden_beit.py): state_dict layout, four output maps, input gradient and digests of all
parameter gradients, eval and train mode.

CPU tier: the gather inside ops.modules is patched with the oracle's torch restatement (the product
has no CPU kernel), everything else that runs is product code.  GPU tier: the HIP kernels, fp32."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbone_cases as bc
from oracle import msda as oracle_msda
from oracle import seeded


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'beit_adapter.npz'))


def _model(name, dev):
    from vitadapter.backbones.beit_adapter import BEiTAdapter
    torch.manual_seed(0)
    model = BEiTAdapter(**bc.BEIT_CASES[name]['cfg'])
    missing, unexpected = model.load_state_dict(seeded.seeded_state_dict(bc.float_shapes(model), 21), strict=False)
    assert not unexpected and all(k.endswith('relative_position_index') for k in missing)
    return model.to(dev)


def _check(model, gold, name, dev, tol_out, tol_gx, tol_gp):
    case = bc.BEIT_CASES[name]
    for mode in case['modes']:
        model.train(mode == 'train')
        model.zero_grad(set_to_none=True)
        x = bc.beit_input(name).to(dev).requires_grad_(True)
        outs = model(x)
        tag = '%s_%s' % (name, mode)
        for k, o in enumerate(outs):
            want = gold['%s_f%d' % (tag, k + 1)]
            assert tuple(o.shape) == want.shape
            err = np.abs(o.detach().cpu().numpy() - want).max()
            assert err <= tol_out * max(1.0, np.abs(want).max()), (tag, k, err)
        gouts = [g.to(dev) for g in bc.beit_gouts(name, [o.shape for o in outs])]
        sum((o * g).sum() for o, g in zip(outs, gouts)).backward()
        want = gold[tag + '_gx']
        err = np.abs(x.grad.cpu().numpy() - want).max()
        assert err <= tol_gx * max(1.0, np.abs(want).max()), (tag, 'gx', err)
        n = 0
        for k, p in model.named_parameters():
            key = '%s_gp_%s' % (tag, k)
            if key in gold.files:
                assert p.grad is not None, key
                w = gold[key]
                assert np.abs(seeded.digest(p.grad.cpu()) - w).max() <= tol_gp * max(1.0, np.abs(w).max()), key
                n += 1
        assert n > 150


@pytest.mark.parametrize('name', sorted(bc.BEIT_CASES))
def test_state_dict_layout_matches_reference(golden_dir, name):
    from vitadapter.backbones.beit_adapter import BEiTAdapter
    meta = json.loads(str(_gold(golden_dir)['meta']))[name]
    sd = BEiTAdapter(**bc.BEIT_CASES[name]['cfg']).state_dict()
    assert sorted(sd) == sorted(meta)
    for k, v in sd.items():
        assert list(v.shape) == meta[k], k


def test_relative_position_index_matches_reference_recipe():
    """beit.py:86-101 for a 3x2 grid, written out by hand."""
    from vitadapter.backbones.beit import relative_position_index
    idx, n_rel = relative_position_index((3, 2))
    assert n_rel == 5 * 3 + 3 and idx.shape == (7, 7)
    assert idx[0, 0] == n_rel - 1 and (idx[0, 1:] == n_rel - 3).all() and (idx[1:, 0] == n_rel - 2).all()
    # token (y, x) -> 1 + 2 y + x;  offset (dy, dx) -> (dy + 2) * 3 + (dx + 1)
    assert idx[1 + 0, 1 + 5] == (0 - 2 + 2) * 3 + (0 - 1 + 1) and idx[1 + 5, 1 + 0] == (2 + 2) * 3 + (1 + 1)
    assert (idx[1:, 1:].diagonal() == 2 * 3 + 1).all()


@pytest.mark.parametrize('name', sorted(bc.BEIT_CASES))
def test_host_logic_cpu(monkeypatch, golden_dir, name):
    import ops.modules.ms_deform_attn as mod

    class _OracleFunction:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return oracle_msda.core_torch(value, shapes, loc, attn)
    monkeypatch.setattr(mod, 'MSDeformAttnFunction', _OracleFunction)
    _check(_model(name, 'cpu'), _gold(golden_dir), name, 'cpu', 5e-5, 1e-4, 5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(bc.BEIT_CASES))
def test_hip_path_fp32(golden_dir, name):
    model = _model(name, 'cuda')
    _check(model, _gold(golden_dir), name, 'cuda', 2e-4, 2e-3, 2e-3)


@pytest.mark.gpu
def test_hip_path_bf16_autocast_runs_and_tracks_fp32():
    """bf16 autocast (fused LayerNorm / residual / Linear / tail kernels + the fused MSDA core): finite,
    and close to the fp32 result at bf16 tolerance."""
    name = 'beit_seg_64'
    model = _model(name, 'cuda').train()
    x = bc.beit_input(name).cuda()
    with torch.no_grad():
        ref = model.eval()(x)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = model(x)
    for o, r in zip(out, ref):
        assert torch.isfinite(o).all()
        assert (o.float() - r).abs().max().item() <= 0.08 * max(1.0, r.abs().max().item())
    model.train()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        outs = model(x.requires_grad_(True))
    sum(o.float().mean() for o in outs).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


@pytest.mark.gpu
def test_hip_path_bf16_bias_attention_gradients_track_fp32():
    """The relative position bias inside the MFMA attention kernels (csrc/attn_flash.hip, bf16 autocast) against the
    fp32 run of the same model: gradients of every relative_position_bias_table (they leave the dQ pass as dS per
    image, are summed over the batch and scattered back through the index) - relative L2 <= 0.15 each, median <= 0.08."""
    import numpy as np
    name = 'beit_seg_64'
    model = _model(name, 'cuda').train()
    for m in model.modules():                                      # deterministic graph for the two runs
        if m.__class__.__name__ == 'DropPath':
            m.drop_prob = 0.
    x = bc.beit_input(name).cuda()
    grads = {}
    gouts = None
    for amp in (False, True):
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            outs = model(x)
        if gouts is None:
            g = torch.Generator(device='cuda').manual_seed(3)
            gouts = [torch.randn(o.shape, device='cuda', generator=g) for o in outs]
        sum((o.float() * go).sum() for o, go in zip(outs, gouts)).backward()
        grads[amp] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters()
                      if 'relative_position_bias_table' in k and p.grad is not None}
    assert len(grads[True]) >= 2 and set(grads[True]) == set(grads[False])
    rels = [float((grads[True][k] - grads[False][k]).norm() / grads[False][k].norm().clamp_min(1e-12)) for k in grads[False]]
    assert max(rels) <= 0.15 and float(np.median(rels)) <= 0.08, rels


@pytest.mark.gpu
def test_beit_large_640_bf16_forward_backward():
    """BASELINE configs[3] as published: BEiT-L adapter (embed 1024, depth 24, 16 heads, relative position bias, class
    token, layer scale 1e-6, deform heads 16, ratio 0.5, with_cp) at 640 x 640, batch 2, train mode, bf16 autocast
    (upernet_beit_adapter_large_640_160k_ade20k_ss.py:13-33): forward + backward through the bias attention kernels
    (N = 1601 tokens), pyramid shapes, finite gradients for every parameter that has one."""
    from vitadapter.backbones.beit_adapter import BEiTAdapter
    torch.manual_seed(0)
    model = BEiTAdapter(img_size=640, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                        use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=1e-6, drop_path_rate=0.3, conv_inplane=64,
                        n_points=4, deform_num_heads=16, cffn_ratio=0.25, deform_ratio=0.5, with_cp=True,
                        interaction_indexes=[[0, 5], [6, 11], [12, 17], [18, 23]]).cuda().train()
    x = torch.randn(2, 3, 640, 640, device='cuda', generator=torch.Generator(device='cuda').manual_seed(2))
    with torch.autocast('cuda', dtype=torch.bfloat16):
        outs = model(x)
    assert [tuple(o.shape) for o in outs] == [(2, 1024, 160, 160), (2, 1024, 80, 80), (2, 1024, 40, 40), (2, 1024, 20, 20)]
    sum(o.float().pow(2).mean() for o in outs).backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert len(grads) > 500 and all(torch.isfinite(g).all() for g in grads.values())
    assert sum('relative_position_bias_table' in k for k in grads) == 24
