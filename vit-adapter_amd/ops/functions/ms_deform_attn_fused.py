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
        with torch.cuda.device(value.device):
            rc = _vah.lib.vah_msda_fused_forward(
                value.data_ptr(), _DT[value.dtype], spatial_shapes.data_ptr(),
                level_start_index.data_ptr(), offsets.data_ptr(), logits.data_ptr(),
                _DT[offsets.dtype], ref.data_ptr(), ref.shape[1], N, S, M, D, L, Lq, P,
                out.data_ptr(), torch.cuda.current_stream(value.device).cuda_stream)
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
        d_off = torch.empty_like(offsets)
        d_logit = torch.empty_like(logits)
        if ctx.tiled:
            # atomic-free tile pass (csrc/msda_tile.hip): grad_value is STORED, in the value's dtype
            sh_host, lsi_host, _ = _vah.host_geometry(shapes, lsi)
            ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P, sh_host, lsi_host)
            if ws_bytes >= 0:
                grad_value = torch.empty_like(value)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
                with torch.cuda.device(value.device):
                    rc = _vah.lib.vah_msda_fused_backward_tiled(
                        value.data_ptr(), _DT[value.dtype], shapes.data_ptr(), lsi.data_ptr(),
                        offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], ref.data_ptr(),
                        ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                        _DT[value.dtype], d_off.data_ptr(), d_logit.data_ptr(), sh_host, lsi_host,
                        ws.data_ptr(), ws_bytes, torch.cuda.current_stream(value.device).cuda_stream)
                _vah.check(rc, 'vah_msda_fused_backward_tiled')
                return grad_value, None, None, d_off, d_logit, None
        # fallback: one float atomic per sample, corner and channel into a zeroed fp32 grad_value
        grad_value = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
        with torch.cuda.device(value.device):
            rc = _vah.lib.vah_msda_fused_backward(
                value.data_ptr(), _DT[value.dtype], shapes.data_ptr(), lsi.data_ptr(),
                offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], ref.data_ptr(),
                ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                d_off.data_ptr(), d_logit.data_ptr(), torch.cuda.current_stream(value.device).cuda_stream)
        _vah.check(rc, 'vah_msda_fused_backward')
        return grad_value.to(value.dtype), None, None, d_off, d_logit, None
