"""CPU: pin the oracle (oracle/msda_oracle.c and oracle/msda.core_torch) against the golden
vectors that tools/gen_golden.py produced from the reference's own Python
(ms_deform_attn_core_pytorch + autograd, detection/ops/functions/ms_deform_attn_func.py:49-71).

Shapes of G1 are the reference's only test, detection/ops/test.py:16-21,108."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, msda, seeded


def _check_digest(stored, tensors):
    got = np.concatenate([seeded.digest(t) for t in tensors])
    np.testing.assert_allclose(got, stored, rtol=1e-12, atol=1e-12,
                               err_msg='seeded inputs differ from the ones the goldens were made with')


def _gate_mask(loc, hw):
    """0 where a sample's pixel coordinate is EXACTLY -1, else 1.

    The one place the reference's two implementations disagree: its CUDA kernel skips a
    sample whose h_im or w_im is exactly -1 (strict gate, ms_deform_im2col_cuda.cuh:288) and
    so writes grad_sampling_loc = 0, while its PyTorch path (grid_sample) returns the one-sided
    derivative there.  Value, grad_value and grad_attn_weight agree (all zero weight).  The
    oracle and the HIP kernels follow the CUDA kernel; the goldens come from the PyTorch path,
    so these measure-zero samples are excluded from the grad_loc comparison ('edge' cases)."""
    norm = np.stack([hw.numpy()[:, 1], hw.numpy()[:, 0]], -1).astype(np.float64)
    px = loc.numpy().astype(np.float64) * norm[None, None, None, :, None, :] - 0.5
    on_gate = (px == -1.0).any(-1, keepdims=True)
    return np.where(on_gate, 0.0, 1.0)


@pytest.fixture(scope='module')
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, 'msda_testpy.npz'))


@pytest.fixture(scope='module')
def g2(golden_dir):
    return np.load(os.path.join(golden_dir, 'msda_adapter.npz'))


@pytest.mark.parametrize('D', cases.TESTPY_CHANNELS)
def test_c_oracle_fp64_matches_reference_testpy(g1, D):
    value, hw, lsi, loc, attn, gout = cases.testpy_inputs(D)
    _check_digest(g1['D%d_digest' % D], (value, loc, attn, gout))
    args = [t.numpy() for t in (value, hw, lsi, loc, attn)]
    out = msda.forward(*args)
    gv, gl, ga = msda.backward(*args, gout.numpy())
    # fp64 pin: the reference test itself uses torch.allclose defaults (rtol 1e-5, atol 1e-8,
    # test.py:40); we hold the oracle to 1e-12 absolute.
    for name, got in (('out', out), ('gv', gv), ('gl', gl), ('ga', ga)):
        ref = g1['D%d_%s' % (D, name)]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-12, name


@pytest.mark.parametrize('name', sorted(cases.ADAPTER_CASES))
def test_c_oracle_matches_reference_adapter_shapes(g2, name):
    value, hw, lsi, loc, attn, gout = cases.msda_inputs(name, **cases.ADAPTER_CASES[name])
    _check_digest(g2[name + '_digest'], (value, loc, attn, gout))
    for dt, tol in ((np.float64, 2e-6), (np.float32, 1e-4)):
        # fp64 oracle vs fp32-rounded fp64 truth: bounded by the fixture's storage rounding
        # (values up to ~30 -> 2e-6); fp32 oracle: the north-star tolerance, 1e-4.
        args = [value.numpy().astype(dt), hw.numpy(), lsi.numpy(), loc.numpy().astype(dt),
                attn.numpy().astype(dt)]
        out = msda.forward(*args)
        gv, gl, ga = msda.backward(*args, gout.numpy().astype(dt))
        for nm, got in (('out', out), ('gv', gv), ('gl', gl), ('ga', ga)):
            ref = g2['%s_%s' % (name, nm)].astype(np.float64)
            scale = max(1.0, np.abs(ref).max())
            err = np.abs(got - ref)
            if nm == 'gl':
                err = err * _gate_mask(loc, hw)
            assert err.max() <= tol * scale, (nm, dt)


@pytest.mark.parametrize('name', ['inj_adapter', 'ext_oob', 'l4_d32', 'd24_p3'])
def test_torch_port_matches_reference(g2, name):
    """core_torch is the CPU-baseline leg of bench.py; it has to be the reference's function."""
    value, hw, lsi, loc, attn, gout = cases.msda_inputs(name, **cases.ADAPTER_CASES[name])
    v = value.double().requires_grad_(True)
    l = loc.double().requires_grad_(True)
    a = attn.double().requires_grad_(True)
    out = msda.core_torch(v, hw, l, a)
    out.backward(gout.double())
    for nm, got in (('out', out), ('gv', v.grad), ('gl', l.grad), ('ga', a.grad)):
        ref = g2['%s_%s' % (name, nm)].astype(np.float64)
        assert np.abs(got.detach().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), nm


def test_edge_gates():
    """Samples exactly on the (-1, H) gate are dropped; exactly-integer coordinates use
    weight 1 on the low corner (cuh:288, :38-45)."""
    value = np.arange(1, 5, dtype=np.float64).reshape(1, 4, 1, 1)      # 2x2 map: [[1,2],[3,4]]
    hw = np.array([[2, 2]], dtype=np.int64)
    lsi = np.array([0], dtype=np.int64)
    attn = np.ones((1, 1, 1, 1, 1))

    def at(x_px, y_px):      # pixel coords -> normalised loc
        loc = np.array([(x_px + 0.5) / 2, (y_px + 0.5) / 2]).reshape(1, 1, 1, 1, 1, 2)
        return msda.forward(value, hw, lsi, loc, attn)[0, 0, 0]

    assert at(0, 0) == 1 and at(1, 0) == 2 and at(0, 1) == 3 and at(1, 1) == 4
    assert at(0.5, 0.5) == 2.5
    assert at(-1, 0) == 0            # w_im == -1 fails the strict gate
    assert at(-0.5, 0) == 0.5        # half outside: zero padding
    assert at(2, 0) == 0             # w_im == W fails the strict gate
    assert at(1.5, 1) == 2.0         # right neighbour out of range -> 0.5 * 4
