"""CPU: the C-ABI library loads and exports every symbol include/vitadapter_hip.h declares;
the Python binding mirrors the reference extension's error behaviour on CPU tensors.
No compute is launched here (there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'vitadapter_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vah_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import _vah
    syms = _declared_symbols()
    assert len(syms) >= 8
    assert sorted(_vah.EXPORTS) == syms
    lib = ctypes.CDLL(_vah.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.vah_abi_version() == _vah.ABI_VERSION


def test_argument_errors_without_gpu():
    """Argument validation happens before anything touches the device."""
    import _vah
    rc = _vah.lib.vah_msda_forward_f32(None, None, None, None, None, 1, 4, -2, 32, 1, 3, 4, None, None)
    assert rc == -2 and b'bad dims' in _vah.lib.vah_last_error()
    rc = _vah.lib.vah_msda_forward_f32(None, None, None, None, None, 1, 4, 2, 32, 1, 3, 4, None, None)
    assert rc == -1 and b'null' in _vah.lib.vah_last_error()
    # zero queries: nothing to do, no pointer is dereferenced
    assert _vah.lib.vah_msda_forward_f32(None, None, None, None, None, 1, 4, 2, 32, 1, 0, 4, None, None) == 0
    with pytest.raises(RuntimeError, match='code -1'):
        _vah.check(-1, 'x')


def test_binding_rejects_cpu_tensors_like_the_reference():
    import MultiScaleDeformableAttention as MSDA
    v = torch.zeros(1, 4, 2, 32)
    s = torch.tensor([[2, 2]])
    i = torch.tensor([0])
    loc = torch.zeros(1, 3, 2, 1, 4, 2)
    att = torch.zeros(1, 3, 2, 1, 4)
    with pytest.raises(RuntimeError, match='Not implemented on the CPU'):
        MSDA.ms_deform_attn_forward(v, s, i, loc, att, 64)
    with pytest.raises(RuntimeError, match='Not implemented on the CPU'):
        MSDA.ms_deform_attn_backward(v, s, i, loc, att, torch.zeros(1, 3, 64), 64)


def test_module_surface_matches_reference():
    """Constructor, parameter names/shapes and initial values of MSDeformAttn
    (detection/ops/modules/ms_deform_attn.py:29-81)."""
    from ops.modules import MSDeformAttn
    m = MSDeformAttn(d_model=768, n_levels=3, n_heads=12, n_points=4, ratio=0.5)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        'sampling_offsets.weight': (288, 768), 'sampling_offsets.bias': (288,),
        'attention_weights.weight': (144, 768), 'attention_weights.bias': (144,),
        'value_proj.weight': (384, 768), 'value_proj.bias': (384,),
        'output_proj.weight': (768, 384), 'output_proj.bias': (768,)}
    assert m.im2col_step == 64
    assert float(sd['sampling_offsets.weight'].abs().max()) == 0
    assert float(sd['attention_weights.weight'].abs().max()) == 0
    b = sd['sampling_offsets.bias'].view(12, 3, 4, 2)
    # head 0 points along +x with magnitude p+1; the pattern repeats over levels
    assert torch.allclose(b[0, :, :, 0], torch.tensor([1., 2., 3., 4.]).expand(3, 4))
    assert torch.allclose(b[0, :, :, 1], torch.zeros(3, 4), atol=1e-6)
    assert torch.allclose(b[3, 0, 1], torch.tensor([0., 2.]), atol=1e-5)     # 90 degrees
    with pytest.raises(ValueError):
        MSDeformAttn(d_model=100, n_heads=8)
