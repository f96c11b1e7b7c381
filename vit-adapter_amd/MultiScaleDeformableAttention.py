"""Drop-in for the reference's compiled extension module ``MultiScaleDeformableAttention``.

The reference's Python does ``import MultiScaleDeformableAttention as MSDA`` and calls
``MSDA.ms_deform_attn_forward`` / ``MSDA.ms_deform_attn_backward``
(/root/reference/detection/ops/functions/ms_deform_attn_func.py:11,25,42; pybind exports
/root/reference/detection/ops/src/vision.cpp:13-16).  This module provides the same two
callables with the same argument order, checks and error behaviour, implemented as a ctypes
binding over the C ABI of libvitadapter_hip.so (include/vitadapter_hip.h).  The GPU work is
hand-written HIP for gfx950; there is no CPU or PyTorch fallback in here.

Checks mirrored from /root/reference/detection/ops/src/cuda/ms_deform_attn_cuda.cu:
  contiguity of every tensor (:28-32, :92-98), device residency (:34-38),
  ``batch % min(batch, im2col_step) == 0`` (:50-52), float/double only (:64,134).
CPU tensors raise "Not implemented on the CPU" as ms_deform_attn.h:38 does.
"""
import os
import threading

import torch

import _vah

__all__ = ['ms_deform_attn_forward', 'ms_deform_attn_backward']


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _common_checks(named, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                   im2col_step):
    if not value.is_cuda:
        raise RuntimeError('Not implemented on the CPU')
    for name, t in named:
        _require(t.is_contiguous(), '%s tensor has to be contiguous' % name)
    for name, t in named:
        _require(t.is_cuda, '%s must be a CUDA tensor' % name)
        _require(t.device == value.device, '%s must be on the same device as value' % name)
    _require(value.dtype in (torch.float32, torch.float64),
             'ms_deform_attn: expected float32 or float64 value, got %s' % value.dtype)
    for name, t in named:
        if name in ('spatial_shapes', 'level_start_index'):
            _require(t.dtype == torch.int64, '%s must be int64' % name)
        else:
            _require(t.dtype == value.dtype, '%s must have the dtype of value' % name)
    _require(value.dim() == 4, 'value must be (N, S, M, D)')
    _require(spatial_shapes.dim() == 2 and spatial_shapes.size(1) == 2,
             'spatial_shapes must be (L, 2)')
    _require(sampling_loc.dim() == 6 and sampling_loc.size(5) == 2,
             'sampling_loc must be (N, Lq, M, L, P, 2)')
    N, S, M, D = value.shape
    L = spatial_shapes.size(0)
    Lq, P = sampling_loc.size(1), sampling_loc.size(4)
    _require(level_start_index.numel() == L, 'level_start_index must be (L,)')
    _require(tuple(sampling_loc.shape) == (N, Lq, M, L, P, 2),
             'sampling_loc shape %s does not match value/spatial_shapes' % (tuple(sampling_loc.shape),))
    _require(tuple(attn_weight.shape) == (N, Lq, M, L, P),
             'attn_weight shape %s does not match sampling_loc' % (tuple(attn_weight.shape),))
    step = min(N, int(im2col_step))
    _require(N == 0 or (step > 0 and N % step == 0),
             'batch(%d) must divide im2col_step(%d)' % (N, step))
    return N, S, M, D, L, Lq, P


def _stream_and_guard(t):
    dev = t.device
    stream = _vah.raw_stream(dev)
    return stream, (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else None)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """-> output (N, Lq, M*D).  Same contract as the reference's ms_deform_attn_forward."""
    named = (('value', value), ('spatial_shapes', spatial_shapes),
             ('level_start_index', level_start_index), ('sampling_loc', sampling_loc),
             ('attn_weight', attn_weight))
    N, S, M, D, L, Lq, P = _common_checks(named, value, spatial_shapes, level_start_index,
                                          sampling_loc, attn_weight, im2col_step)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    fn = _vah.lib.vah_msda_forward_f32 if value.dtype == torch.float32 else _vah.lib.vah_msda_forward_f64
    stream, guard = _stream_and_guard(value)
    if guard is not None:
        guard.__enter__()
    try:
        rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P,
                out.data_ptr(), stream)
    finally:
        if guard is not None:
            guard.__exit__(None, None, None)
    _vah.check(rc, 'ms_deform_attn_forward')
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (shapes of the inputs)."""
    named = (('value', value), ('spatial_shapes', spatial_shapes),
             ('level_start_index', level_start_index), ('sampling_loc', sampling_loc),
             ('attn_weight', attn_weight), ('grad_output', grad_output))
    N, S, M, D, L, Lq, P = _common_checks(named, value, spatial_shapes, level_start_index,
                                          sampling_loc, attn_weight, im2col_step)
    _require(grad_output.numel() == N * Lq * M * D, 'grad_output must be (N, Lq, M*D)')
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    fn = _vah.lib.vah_msda_backward_f32 if value.dtype == torch.float32 else _vah.lib.vah_msda_backward_f64
    # atomic-free tiled grad_value (csrc/msda_tile.hip): fp32, D == 32, P == 4, L <= 4.  Grid and workspace follow
    # from the tensor SHAPES alone; the level geometry is read on the device (no D2H read, no host cache)
    tiled = None
    if (value.dtype == torch.float32 and D == 32 and P == 4 and 1 <= L <= 4 and N * Lq * M > 0
            and os.environ.get('VAH_MSDA_TILED', '1') != '0'):
        ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P)
        if ws_bytes >= 0:
            tiled = (torch.empty(ws_bytes, dtype=torch.uint8, device=value.device), ws_bytes)
    # the tile pass stores every element of grad_value; the scatter kernels accumulate into zeros
    grad_value = torch.empty_like(value) if tiled else torch.zeros_like(value)
    stream, guard = _stream_and_guard(value)
    if guard is not None:
        guard.__enter__()
    try:
        if tiled:
            rc = _vah.lib.vah_msda_backward_tiled_f32(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                tiled[0].data_ptr(), tiled[1], stream)
        else:
            rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                    sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                    N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(),
                    grad_attn.data_ptr(), stream)
    finally:
        if guard is not None:
            guard.__exit__(None, None, None)
    _vah.check(rc, 'ms_deform_attn_backward')
    return [grad_value, grad_loc, grad_attn]
