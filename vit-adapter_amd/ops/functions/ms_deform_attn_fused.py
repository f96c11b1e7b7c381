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


class WindowSchedule:
    """Query groups + value windows for the LDS-window forward (include/vitadapter_hip.h,
    vah_msda_fused_forward_win).  Static per (reference grid, level geometry)."""

    def __init__(self, perm, group_off, group_win, ngroups, max_win_px, H, W, start):
        self.perm, self.group_off, self.group_win = perm, group_off, group_win
        self.ngroups, self.max_win_px, self.H, self.W, self.start = int(ngroups), int(max_win_px), int(H), int(W), int(start)


_WIN_CACHE = {}


def build_window_schedule(reference_points, H, W, start, tile=8, halo=None):
    """Queries grouped by the ``tile`` x ``tile``-pixel tile of the (single) value map their reference point falls
    in; the group's window is the tile + ``halo`` + 1 pixels on every side, clipped to the map (a sample whose
    offset stays within ``halo`` pixels has all four corners inside; the others read global memory)."""
    halo = int(os.environ.get('VAH_MSDA_WIN_HALO', 5)) if halo is None else int(halo)
    ref = reference_points.detach().float().reshape(-1, reference_points.shape[-2], 2)[:, 0]       # (Lq, 2) as (x, y)
    ty = (ref[:, 1] * H / tile).floor().clamp_(0, (H - 1) // tile).long()
    tx = (ref[:, 0] * W / tile).floor().clamp_(0, (W - 1) // tile).long()
    ntx, nty = (W + tile - 1) // tile, (H + tile - 1) // tile
    gid = ty * ntx + tx
    perm = torch.argsort(gid, stable=True).to(torch.int32)
    counts = torch.bincount(gid, minlength=nty * ntx)
    group_off = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
    wins, max_px = [], 1
    for gy in range(nty):
        for gx in range(ntx):
            y0, x0 = max(gy * tile - halo - 1, 0), max(gx * tile - halo - 1, 0)
            y1, x1 = min(gy * tile + tile + halo + 1, H), min(gx * tile + tile + halo + 1, W)
            wins.append([y0, x0, y1 - y0, x1 - x0])
            max_px = max(max_px, (y1 - y0) * (x1 - x0))
    dev = reference_points.device
    return WindowSchedule(perm.contiguous(), group_off.contiguous(),
                          torch.tensor(wins, dtype=torch.int32, device=dev).contiguous(), nty * ntx, max_px, H, W, start)


def window_schedule_for(reference_points, spatial_shapes, level_start_index, value_dtype):
    """Cached per reference-point tensor.  Single-level calls with batch-shared reference points only
    (VAH_MSDA_FWD_WIN=0: the 8-lane gather kernel)."""
    if os.environ.get('VAH_MSDA_FWD_WIN', '1') == '0' or spatial_shapes.shape[0] != 1:
        return None
    key = (reference_points.data_ptr(), tuple(reference_points.shape), reference_points._version,
           spatial_shapes.data_ptr(), str(reference_points.device))
    hit = _WIN_CACHE.get(key)
    if hit is None:
        _, _, sh = _vah.host_geometry(spatial_shapes, level_start_index)
        start = int(level_start_index.tolist()[0])
        sched = build_window_schedule(reference_points, sh[0], sh[1], start)
        if sched.max_win_px * 32 * (2 if value_dtype == torch.bfloat16 else 4) > 64 * 1024:
            sched = None
        if len(_WIN_CACHE) > 64:
            _WIN_CACHE.clear()
        hit = (sched, reference_points, spatial_shapes)         # keep the keys' tensors alive
        _WIN_CACHE[key] = hit
    return hit[0]


def tiled_backward(n_levels, n_points):
    """The atomic-free tile pass serves P == 4, L <= 4 (VAH_MSDA_TILED=0: per-sample float atomics, for A/B runs)."""
    return n_points == 4 and 1 <= n_levels <= 4 and os.environ.get('VAH_MSDA_TILED', '1') != '0'


class MSDeformAttnFusedFunction(Function):
    """apply(value (N,S,M,32), spatial_shapes, level_start_index, offsets (N,Lq,M,L,P,2),
    logits (N,Lq,M,L*P), reference_points (1,Lq,1|L,2)) -> (N, Lq, M*32) in value's dtype."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, offsets, logits, reference_points):
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = offsets.shape
        value, offsets, logits = value.contiguous(), offsets.contiguous(), logits.contiguous()
        ref = reference_points.detach().float().contiguous().view(Lq, -1, 2)
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        win = window_schedule_for(reference_points, spatial_shapes, level_start_index, value.dtype) \
            if (L == 1 and P == 4 and ref.shape[1] == 1) else None
        if win is not None:
            with _vah.on(value.device):
                rc = _vah.lib.vah_msda_fused_forward_win(
                    value.data_ptr(), _DT[value.dtype], offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype],
                    ref.data_ptr(), win.perm.data_ptr(), win.group_off.data_ptr(), win.group_win.data_ptr(),
                    win.ngroups, win.max_win_px, win.H, win.W, win.start, N, S, M, D, Lq, P, out.data_ptr(),
                    _vah.raw_stream(value.device))
            _vah.check(rc, 'vah_msda_fused_forward_win')
            ctx.save_for_backward(value, spatial_shapes, level_start_index, offsets, logits, ref)
            ctx.tiled = tiled_backward(L, P)
            return out
        with _vah.on(value.device):
            rc = _vah.lib.vah_msda_fused_forward(
                value.data_ptr(), _DT[value.dtype], spatial_shapes.data_ptr(),
                level_start_index.data_ptr(), offsets.data_ptr(), logits.data_ptr(),
                _DT[offsets.dtype], ref.data_ptr(), ref.shape[1], N, S, M, D, L, Lq, P,
                out.data_ptr(), _vah.raw_stream(value.device))
        _vah.check(rc, 'vah_msda_fused_forward')
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
                        offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], ref.data_ptr(),
                        ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                        _DT[value.dtype], d_off.data_ptr(), d_logit.data_ptr(), _DT[gdt],
                        ws.data_ptr(), ws_bytes, _vah.raw_stream(value.device))
                _vah.check(rc, 'vah_msda_fused_backward_tiled')
                return grad_value, None, None, d_off, d_logit, None
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
