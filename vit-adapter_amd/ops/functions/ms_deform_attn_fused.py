"""Fused core of MSDeformAttn.forward: softmax(logits) + sampling locations + gather in one HIP
kernel (csrc/msda_fused.hip); gradients by the tile pass of csrc/msda_tile.hip.  Not part of the reference's surface (its
module does these steps with ~8 PyTorch ops around MSDeformAttnFunction,
/root/reference/detection/ops/modules/ms_deform_attn.py:108-128); ops.modules.MSDeformAttn uses it
when the shapes allow and otherwise keeps the reference sequence.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import _vah

_DT = {torch.float32: 0, torch.bfloat16: 1}


def fused_supported(value, offsets, logits, reference_points, n_levels, n_points):
    """True when the fused kernels cover this call (else use the unfused Function)."""
    if os.environ.get('VAH_MSDA_FUSED', '1') == '0':
        return False
    return (value.is_cuda and value.dim() == 4 and value.dtype in _DT and offsets.dtype in _DT
            and logits.dtype == offsets.dtype and reference_points.shape[0] == 1
            and reference_points.shape[-1] == 2 and reference_points.shape[2] in (1, n_levels)
            and bool(_vah.lib.vah_msda_fused_supported(value.shape[-1], n_levels, n_points))
            and value.numel() > 0 and offsets.numel() > 0)


WIN_HALO = int(os.environ.get('VAH_MSDA_WIN_HALO', 5))


def window_forward(n_levels, n_points, ref_levels, Lq):
    """The LDS-window forward serves single-level calls with batch-shared reference points (VAH_MSDA_FWD_WIN=0: the
    8-lane gather kernel).  Its schedule is built on the device from the reference points and the device copies of the
    level geometry: nothing is read back (see _window_workspace for how one forward's six calls share it)."""
    return (n_levels == 1 and n_points == 4 and ref_levels == 1 and Lq <= (1 << 18)
            and os.environ.get('VAH_MSDA_FWD_WIN', '1') != '0')


def _row_strides(offsets, logits):
    """(offsets', logits', os, ls): element strides between consecutive (n, q, m) rows when both tensors are laid out row
    by row with a constant row stride and contiguous rows (contiguous tensors, or views into the module's interleaved
    [offsets | logits] matrix); contiguous copies otherwise."""
    def rows(t, inner):
        st = t.stride()
        shp = t.shape
        if t.is_contiguous():
            return inner
        # inner block contiguous, (n, q, m) collapse to one row index with a constant stride
        blk = 1
        for d in range(t.dim() - 1, 2, -1):
            if st[d] != blk:
                return None
            blk *= shp[d]
        rs = st[2]
        if rs < inner or st[1] != shp[2] * rs or st[0] != shp[1] * shp[2] * rs:
            return None
        return rs
    N, Lq, M, L, P, _ = offsets.shape
    o_s, l_s = rows(offsets, L * P * 2), rows(logits, L * P)
    if o_s is None or l_s is None:
        return offsets.contiguous(), logits.contiguous(), L * P * 2, L * P
    return offsets, logits, o_s, l_s


def tiled_backward(n_levels, n_points):
    """The atomic-free tile pass serves P == 4, L <= 4 (VAH_MSDA_TILED=0: per-sample float atomics, for A/B runs)."""
    return n_points == 4 and 1 <= n_levels <= 4 and os.environ.get('VAH_MSDA_TILED', '1') != '0'


class _WinSchedule:
    """A window-forward workspace that still holds its schedule, riding on the caller's reference_points tensor."""
    __slots__ = ('ws', 'key', 'shapes', 'lsi')


def _window_workspace(carrier, token, spatial_shapes, level_start_index, S, Lq, device):
    """-> (ws, ws_bytes, holds_schedule).  The schedule of the window forward depends on the reference points and the level
    geometry only, and the adapter calls its extractor six times per forward (and six more in a checkpointed backward)
    with ONE set of deform inputs: the workspace rides on the reference_points tensor OBJECT the caller passed
    (`carrier`) and is handed to the next call that comes with the same objects, unmodified (tensor version counters),
    on the same stream, in the same pass (`token`: the caller's forward epoch + capture state).  No table keyed by id():
    the attribute dies with the tensor, a fresh tensor simply has none; nothing is read back from the device."""
    import weakref
    ws_bytes = _vah.lib.vah_msda_win_ws_bytes(S, Lq)
    if ws_bytes < 0:
        return None, ws_bytes, False
    key = None
    if carrier is not None and token is not None:
        key = (token, carrier._version, spatial_shapes._version, level_start_index._version, S, Lq, ws_bytes,
               _vah.raw_stream(device), torch.cuda.is_current_stream_capturing())
        sch = getattr(carrier, '_vah_win_schedule', None)
        if (sch is not None and sch.key == key and sch.shapes() is spatial_shapes and sch.lsi() is level_start_index
                and sch.ws.device == device):
            return sch.ws, ws_bytes, True
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    if key is not None:
        sch = _WinSchedule()
        sch.ws, sch.key = ws, key
        sch.shapes, sch.lsi = weakref.ref(spatial_shapes), weakref.ref(level_start_index)
        carrier._vah_win_schedule = sch
    return ws, ws_bytes, False


def fused_forward(value, spatial_shapes, level_start_index, offsets, logits, o_s, l_s, ref, carrier=None, token=None):
    """The fused forward kernels on raw row-strided offsets / logits (see _row_strides); ref (Lq, 1 | L, 2) fp32.
    carrier / token: see _window_workspace (None: the window schedule is built in this call)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = offsets.shape
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    if window_forward(L, P, ref.shape[1], Lq):
        ws, ws_bytes, ready = _window_workspace(carrier, token, spatial_shapes, level_start_index, S, Lq, value.device)
        if ws is not None:
            with _vah.on(value.device):
                rc = _vah.lib.vah_msda_fused_forward_win(
                    value.data_ptr(), _DT[value.dtype], spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                    offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], o_s, l_s, ref.data_ptr(),
                    N, S, M, D, Lq, P, WIN_HALO, ws.data_ptr(), ws_bytes, int(ready), out.data_ptr(), _vah.raw_stream(value.device))
            _vah.check(rc, 'vah_msda_fused_forward_win')
            return out
    with _vah.on(value.device):
        rc = _vah.lib.vah_msda_fused_forward(
            value.data_ptr(), _DT[value.dtype], spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), offsets.data_ptr(), logits.data_ptr(),
            _DT[offsets.dtype], o_s, l_s, ref.data_ptr(), ref.shape[1], N, S, M, D, L, Lq, P,
            out.data_ptr(), _vah.raw_stream(value.device))
    _vah.check(rc, 'vah_msda_fused_forward')
    return out


class MSDeformAttnFusedFunction(Function):
    """apply(value (N,S,M,32), spatial_shapes, level_start_index, offsets (N,Lq,M,L,P,2),
    logits (N,Lq,M,L*P), reference_points (1,Lq,1|L,2)) -> (N, Lq, M*32) in value's dtype."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, offsets, logits, reference_points):
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = offsets.shape
        value = value.contiguous()
        offsets, logits, o_s, l_s = _row_strides(offsets, logits)
        ref = reference_points.detach().float().contiguous().view(Lq, -1, 2)
        out = fused_forward(value, spatial_shapes, level_start_index, offsets, logits, o_s, l_s, ref)
        ctx.save_for_backward(value, spatial_shapes, level_start_index, offsets, logits, ref)
        ctx.tiled = tiled_backward(L, P)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, offsets, logits, ref = ctx.saved_tensors
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = offsets.shape
        grad_output = grad_output.contiguous().to(value.dtype)
        offsets, logits, o_s, l_s = _row_strides(offsets, logits)
        if ctx.tiled:
            # atomic-free tile pass (csrc/msda_tile.hip): grad_value is STORED, in the value's dtype.  Nothing about the
            # level geometry is read back to the host: grid and workspace follow from the tensor shapes.
            ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P)
            if ws_bytes >= 0:
                gdt = offsets.dtype
                d_off = torch.empty(offsets.shape, dtype=gdt, device=offsets.device)
                d_logit = torch.empty(logits.shape, dtype=gdt, device=logits.device)
                grad_value = torch.empty_like(value)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
                with _vah.on(value.device):
                    rc = _vah.lib.vah_msda_fused_backward_tiled(
                        value.data_ptr(), _DT[value.dtype], shapes.data_ptr(), lsi.data_ptr(),
                        offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], o_s, l_s, ref.data_ptr(),
                        ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                        _DT[value.dtype], d_off.data_ptr(), d_logit.data_ptr(), _DT[gdt], 0, 0,
                        ws.data_ptr(), ws_bytes, _vah.raw_stream(value.device))
                _vah.check(rc, 'vah_msda_fused_backward_tiled')
                return grad_value, None, None, d_off, d_logit, None
        offsets, logits = offsets.contiguous(), logits.contiguous()
        d_off = torch.empty_like(offsets)
        d_logit = torch.empty_like(logits)
        # fallback: one float atomic per sample, corner and channel into a zeroed fp32 grad_value
        grad_value = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
        with _vah.on(value.device):
            rc = _vah.lib.vah_msda_fused_backward(
                value.data_ptr(), _DT[value.dtype], shapes.data_ptr(), lsi.data_ptr(),
                offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], ref.data_ptr(),
                ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                d_off.data_ptr(), d_logit.data_ptr(), _vah.raw_stream(value.device))
        _vah.check(rc, 'vah_msda_fused_backward')
        return grad_value.to(value.dtype), None, None, d_off, d_logit, None
