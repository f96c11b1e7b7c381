"""BEiTAdapter, DETECTION flavour (no class token, windowed blocks whose grid is zero-padded
BEFORE the projection, global blocks over a window_size^2 grid, per-block (2w-1)^2 relative position tables) against
goldens produced by the reference's own class (tools/gen_golden_beit_det.py from
detection/mmdet_custom/models/backbones/{beit_adapter.py, base/beit.py}): state_dict layout, four output maps, input
gradient and digests of all parameter gradients, eval and train mode.

CPU tier: the gather inside ops.modules is patched with the oracle's torch restatement (the product has no CPU
kernel), everything else that runs is product code.  GPU tier: the HIP kernels in fp32, and bf16 autocast with
head_dim 64 so that the windows run through the bias attention kernels (csrc/attn_flash.hip)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbone_cases as bc
from oracle import msda as oracle_msda
from oracle import seeded


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'beit_adapter_det.npz'))


def _model(name, dev):
    from vitadapter.backbones.beit_det import BEiTAdapter
    torch.manual_seed(0)
    model = BEiTAdapter(**bc.BEIT_DET_CASES[name]['cfg'])
    missing, unexpected = model.load_state_dict(seeded.seeded_state_dict(bc.float_shapes(model), 23), strict=False)
    assert not unexpected and all(k.endswith('relative_position_index') for k in missing)
    return model.to(dev)


def _check(model, gold, name, dev, tol_out, tol_gx, tol_gp):
    case = bc.BEIT_DET_CASES[name]
    for mode in case['modes']:
        model.train(mode == 'train')
        model.zero_grad(set_to_none=True)
        x = bc.beit_det_input(name).to(dev).requires_grad_(True)
        outs = model(x)
        tag = '%s_%s' % (name, mode)
        for k, o in enumerate(outs):
            want = gold['%s_f%d' % (tag, k + 1)]
            assert tuple(o.shape) == want.shape
            err = np.abs(o.detach().cpu().numpy() - want).max()
            assert err <= tol_out * max(1.0, np.abs(want).max()), (tag, k, err)
        gouts = [g.to(dev) for g in bc.beit_det_gouts(name, [o.shape for o in outs])]
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


@pytest.mark.parametrize('name', sorted(bc.BEIT_DET_CASES))
def test_state_dict_layout_matches_reference(golden_dir, name):
    from vitadapter.backbones.beit_det import BEiTAdapter
    meta = json.loads(str(_gold(golden_dir)['meta']))[name]
    sd = BEiTAdapter(**bc.BEIT_DET_CASES[name]['cfg']).state_dict()
    assert sorted(sd) == sorted(meta)
    for k, v in sd.items():
        assert list(v.shape) == meta[k], k


def test_window_relative_position_index():
    """base/beit.py:122-134 for a 3 x 3 window, by hand: offset (dy, dx) -> (dy + 2) * 5 + (dx + 2)."""
    from vitadapter.backbones.beit_det import window_relative_position_index
    idx = window_relative_position_index(3)
    assert idx.shape == (9, 9) and (idx.diagonal() == 2 * 5 + 2).all()
    assert idx[0, 8] == (0 - 2 + 2) * 5 + (0 - 2 + 2) and idx[8, 0] == (2 + 2) * 5 + (2 + 2) and idx[1, 3] == (0 - 1 + 2) * 5 + (1 - 0 + 2)


@pytest.mark.parametrize('name', sorted(bc.BEIT_DET_CASES))
def test_host_logic_cpu(monkeypatch, golden_dir, name):
    import ops.modules.ms_deform_attn as mod

    class _OracleFunction:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return oracle_msda.core_torch(value, shapes, loc, attn)
    monkeypatch.setattr(mod, 'MSDeformAttnFunction', _OracleFunction)
    _check(_model(name, 'cpu'), _gold(golden_dir), name, 'cpu', 5e-5, 1e-4, 5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(bc.BEIT_DET_CASES))
def test_hip_path_fp32(golden_dir, name):
    _check(_model(name, 'cuda'), _gold(golden_dir), name, 'cuda', 2e-4, 2e-3, 2e-3)


@pytest.mark.gpu
def test_hip_path_bf16_windows_on_the_bias_kernels():
    """head_dim 64, a 10 x 8 token grid (padded to 12 x 8 by the 4 x 4 windows), windowed blocks only: bf16 autocast (the windows are one batch of 16-token
    sequences for csrc/attn_flash.hip's bias kernels) against the fp32 run of the same model - outputs within 8e-2 of
    the max, finite gradients, relative position table gradients within 0.15 relative L2."""
    from vitadapter.backbones.beit_det import BEiTAdapter
    torch.manual_seed(0)
    model = BEiTAdapter(img_size=160, patch_size=16, embed_dim=128, depth=4, num_heads=2, mlp_ratio=2, qkv_bias=True,
                        use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=0.1, drop_path_rate=0., conv_inplane=16,
                        n_points=4, deform_num_heads=4, cffn_ratio=0.25, deform_ratio=1.0, with_cp=False,
                        window_attn=[True] * 4, window_size=[4] * 4,
                        interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]]).cuda().train()
    with torch.no_grad():
        for k, p in model.named_parameters():
            if 'relative_position_bias_table' in k:
                p.normal_(0, 0.5)
    x = torch.randn(2, 3, 160, 128, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
    grads, outs_ref, gouts = {}, None, None
    for amp in (False, True):
        model.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            outs = model(x)
        if gouts is None:
            g = torch.Generator(device='cuda').manual_seed(5)
            gouts = [torch.randn(o.shape, device='cuda', generator=g) for o in outs]
            outs_ref = [o.detach().float() for o in outs]
        else:
            for o, r in zip(outs, outs_ref):
                assert torch.isfinite(o).all()
                assert float((o.detach().float() - r).abs().max()) <= 8e-2 * max(1.0, float(r.abs().max()))
        sum((o.float() * go).sum() for o, go in zip(outs, gouts)).backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        grads[amp] = {k: p.grad.detach().double().clone() for k, p in model.named_parameters()
                      if 'relative_position_bias_table' in k and p.grad is not None}
    assert len(grads[True]) == 4
    rels = [float((grads[True][k] - grads[False][k]).norm() / grads[False][k].norm().clamp_min(1e-12)) for k in grads[False]]
    assert max(rels) <= 0.15, rels
