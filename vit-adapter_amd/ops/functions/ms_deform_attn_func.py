"""Autograd entry of multi-scale deformable attention.

Mirror of /root/reference/detection/ops/functions/ms_deform_attn_func.py:19-46: same
``apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
im2col_step)`` signature, inputs cast to fp32 under autocast, once-differentiable backward
returning ``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.
The compute is the HIP library behind ``MultiScaleDeformableAttention``.
"""
import MultiScaleDeformableAttention as MSDA
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

try:                                    # torch >= 2.4 spelling
    from torch.amp import custom_bwd as _custom_bwd, custom_fwd as _custom_fwd

    def custom_fwd(**kw):
        return _custom_fwd(device_type='cuda', **kw)

    def custom_bwd(fn):
        return _custom_bwd(fn, device_type='cuda')
except ImportError:                     # pragma: no cover
    from torch.cuda.amp import custom_bwd, custom_fwd


class MSDeformAttnFunction(Function):
    @staticmethod
    @custom_fwd(cast_inputs=torch.float32)
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations,
            attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    @custom_bwd
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(
            value, shapes, lsi, loc, attn, grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_attn, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations,
                                attention_weights):
    """Pure-PyTorch statement of the op, kept for API parity with the reference
    (ms_deform_attn_func.py:49-71, "for debug and test only").  MSDeformAttnFunction never
    calls it; it is not a fallback."""
    N, _, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    hw = [(int(h), int(w)) for h, w in value_spatial_shapes]
    levels = value.split([h * w for h, w in hw], dim=1)
    grids = 2 * sampling_locations - 1
    taps = []
    for i, (h, w) in enumerate(hw):
        v = levels[i].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        g = grids[:, :, :, i].transpose(1, 2).flatten(0, 1)
        taps.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                  align_corners=False))
    a = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(taps, dim=-2).flatten(-2) * a).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
