"""oracle/msda.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU checkers for multi-scale deformable attention.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; nothing under ``vit-adapter_amd/`` does.

Two restatements live here:

* ``forward`` / ``backward``: ctypes front-end of ``libmsda_oracle.so`` (plain C,
  ``msda_oracle.c``), the scalar statement of
  ``detection/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159,237-403``.
* ``core_torch``: the op-for-op torch sequence of the reference's
  ``ms_deform_attn_core_pytorch`` (``detection/ops/functions/ms_deform_attn_func.py:49-71``):
  split per level -> ``F.grid_sample(bilinear, zeros, align_corners=False)`` -> stack ->
  weight -> sum.  This is what ``bench.py`` times on the host cores as ``cpu_baseline``
  (kind ``"port"``) and what autograd differentiates to give CPU reference gradients.

Both are pinned against ``tests/golden/msda_*.npz`` (made by ``tools/gen_golden.py`` from the
reference's own Python) in ``tests/test_oracle_golden.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile libmsda_oracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, 'libmsda_oracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('msda_oracle.c', 'msda_oracle_body.inc')]
    if force or not os.path.exists(so) or any(
            os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libmsda_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, attn):
    dt = value.dtype
    assert dt in (np.float32, np.float64), dt
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and lsi.shape == (L,)
    assert attn.shape == (N, Lq, M, L, P)
    dims = [ctypes.c_int64(x) for x in (N, S, M, D, L, Lq, P)]
    sfx = '_f32' if dt == np.float32 else '_f64'
    return value, shapes, lsi, loc, attn, dims, sfx, (N, S, M, D, L, Lq, P)


def level_start_index(shapes):
    shapes = np.asarray(shapes, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]]).astype(np.int64)


def forward(value, shapes, lsi, loc, attn):
    """numpy in / numpy out: (N,S,M,D),(L,2),(L,),(N,Lq,M,L,P,2),(N,Lq,M,L,P) -> (N,Lq,M*D)."""
    value, shapes, lsi, loc, attn, dims, sfx, (N, S, M, D, L, Lq, P) = _prep(
        value, shapes, lsi, loc, attn)
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_lib(), 'msda_oracle_forward' + sfx)
    rc = fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn), *dims, _ptr(out))
    assert rc == 0
    return out


def backward(value, shapes, lsi, loc, attn, grad_out):
    """-> (grad_value, grad_loc, grad_attn), shapes of value / loc / attn."""
    value, shapes, lsi, loc, attn, dims, sfx, (N, S, M, D, L, Lq, P) = _prep(
        value, shapes, lsi, loc, attn)
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    assert grad_out.shape == (N, Lq, M * D)
    gv = np.zeros_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_lib(), 'msda_oracle_backward' + sfx)
    rc = fn(_ptr(value), _ptr(shapes), _ptr(lsi), _ptr(loc), _ptr(attn), _ptr(grad_out),
            *dims, _ptr(gv), _ptr(gl), _ptr(ga))
    assert rc == 0
    return gv, gl, ga


def core_torch(value, spatial_shapes, sampling_locations, attention_weights):
    """Torch restatement of the reference's pure-PyTorch path (see module docstring).

    ``spatial_shapes`` may be a tensor or a list of (H, W)."""
    import torch
    import torch.nn.functional as F
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    hw = [(int(h), int(w)) for h, w in (spatial_shapes.tolist()
                                         if hasattr(spatial_shapes, 'tolist') else spatial_shapes)]
    per_level = value.split([h * w for h, w in hw], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        v = per_level[lvl].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))
    a = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
