"""CPU: host-side logic of the product backbone (module wiring, window partitioning, pos-embed
resize matrix, geometry cache, pyramid assembly) against the reference goldens.

There is no GPU in this tier and the product has no CPU path for deformable attention (it
raises "Not implemented on the CPU", as the reference does).  So for THIS TEST ONLY the autograd
function inside ops.modules is monkeypatched with the oracle's torch restatement; everything
else that runs is product code.  The HIP kernels themselves are covered by the -m gpu tier."""
import os

import numpy as np
import pytest
import torch

from oracle import backbone_cases as bc
from oracle import msda as oracle_msda
from oracle import seeded


class _OracleFunction:
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, step):
        return oracle_msda.core_torch(value, shapes, loc, attn)


@pytest.fixture()
def product_with_oracle_msda(monkeypatch):
    import ops.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, 'MSDeformAttnFunction', _OracleFunction)
    from vitadapter.backbones import ViTAdapter
    return ViTAdapter


@pytest.mark.parametrize('name', sorted(bc.FULL_CASES))
def test_product_host_logic_matches_reference(product_with_oracle_msda, golden_dir, name):
    gold = np.load(os.path.join(golden_dir, 'backbone.npz'))
    case = bc.FULL_CASES[name]
    model = product_with_oracle_msda(**case['cfg'])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded.seeded_state_dict(shapes, 5))
    for mode in case['modes']:
        model.train(mode == 'train')
        model.zero_grad(set_to_none=True)
        x = bc.full_input(name).requires_grad_(True)
        outs = model(x)
        tag = '%s_%s' % (name, mode)
        for k, o in enumerate(outs):
            want = gold['%s_f%d' % (tag, k + 1)]
            assert np.abs(o.detach().numpy() - want).max() <= 5e-5 * max(1.0, np.abs(want).max())
        gouts = bc.full_gouts(name, [o.shape for o in outs])
        sum((o * g).sum() for o, g in zip(outs, gouts)).backward()
        want = gold[tag + '_gx']
        assert np.abs(x.grad.numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
        n = 0
        for k, p in model.named_parameters():
            key = '%s_gp_%s' % (tag, k)
            if key in gold.files and p.grad is not None:
                w = gold[key]
                assert np.abs(seeded.digest(p.grad) - w).max() <= 5e-4 * max(1.0, np.abs(w).max()), key
                n += 1
        assert n > 100


def test_product_msda_has_no_cpu_path():
    """Unpatched, the product refuses CPU tensors exactly like the reference extension."""
    from vitadapter.backbones import ViTAdapter
    case = bc.FULL_CASES['seg_glob_64']
    model = ViTAdapter(**case['cfg']).eval()
    with pytest.raises(RuntimeError, match='Not implemented on the CPU'):
        model(bc.full_input('seg_glob_64'))


def test_geometry_cache_matches_reference_recipe():
    """deform_inputs: injector = ViT grid querying the 3-level pyramid, extractor the reverse
    (segmentation/.../adapter_modules.py:28-47)."""
    from vitadapter.backbones.adapter_modules import deform_inputs
    from oracle import cases
    x = torch.zeros(1, 3, 64, 96)
    d1, d2 = deform_inputs(x)
    assert d1[1].tolist() == [[8, 12], [4, 6], [2, 3]] and d1[2].tolist() == [0, 96, 120]
    assert d2[1].tolist() == [[4, 6]] and d2[2].tolist() == [0]
    assert torch.allclose(d1[0], cases.reference_grid([(4, 6)]))
    assert torch.allclose(d2[0], cases.reference_grid([(8, 12), (4, 6), (2, 3)]))
    assert deform_inputs(x)[0][0] is d1[0]          # cached per (H, W, device)
