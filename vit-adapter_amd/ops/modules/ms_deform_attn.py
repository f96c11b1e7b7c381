"""MSDeformAttn module on the gfx950 HIP kernels.

Interface mirror of /root/reference/detection/ops/modules/ms_deform_attn.py:28-130: same
constructor arguments, same four Linear sub-modules with the same names (state_dict keys
``sampling_offsets.*``, ``attention_weights.*``, ``value_proj.*``, ``output_proj.*``), same
``_reset_parameters`` initial values and the same ``forward`` signature and result.

Difference kept on purpose: the reference compares a device scalar with a Python int on every
call (``assert (shapes[:,0]*shapes[:,1]).sum() == Len_in``, :99-100), which costs a GPU->CPU
sync 10x per backbone forward.  Here that check runs only when ``validate_shapes`` is set;
the kernels themselves ignore any level that is not a valid window of the value rows.
"""
import math
import warnings

import MultiScaleDeformableAttention as MSDA
import torch
import torch.nn.functional as F
from torch import nn

from ..functions import MSDeformAttnFunction, MSDeformAttnFusedFunction, fused_supported

# Fused nn.Linear (bf16 working copies, fp32 weight gradients) and the one-node offsets / weights pair + core when the
# backbone package is there.  Resolved at the first forward, not at import: `vitadapter` imports this package, so an import
# here succeeds or fails depending on which of the two is imported first.
_FUSED = None


def _fused():
    global _FUSED
    if _FUSED is None:
        try:
            from vitadapter import fused as mod
            _FUSED = mod
        except ImportError:                                             # ops/ used on its own
            _FUSED = False
    return _FUSED


def _linear(lin, x):
    f = _fused()
    return f.linear(lin, x) if f else lin(x)


def _linear_pair(lin_a, lin_b, x):
    f = _fused()
    return f.linear_pair(lin_a, lin_b, x, f32_out=True) if f else (lin_a(x), lin_b(x))


def _pair_core_ok(mod, query, value, reference_points):
    f = _fused()
    return bool(f) and f.msda_pair_core_ok(mod, query, value, reference_points)


def _pair_core(*args):
    return _fused().msda_pair_core(*args)


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError('invalid input for _is_power_of_2: {} (type: {})'.format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    validate_shapes = False       # set True to get the reference's (syncing) length assertion

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, ratio=1.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, '
                             'but got {} and {}'.format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn('MSDeformAttn: a per-head dimension that is a power of 2 '
                          '(ideally 32) runs on the fast gfx950 kernels.')
        self.im2col_step = 64
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.ratio = ratio
        d_value = int(d_model * ratio)
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_value)
        self.output_proj = nn.Linear(d_value, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """Initial values of the reference (ms_deform_attn.py:62-81): zero offset / weight
        matrices, offset bias = unit ring direction of each head scaled by (point index + 1),
        Xavier value / output projections with zero bias."""
        M, L, P = self.n_heads, self.n_levels, self.n_points
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            angle = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
            ring = torch.stack([angle.cos(), angle.sin()], -1)
            ring = ring / ring.abs().max(-1, keepdim=True)[0]
            scale = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)
            bias = ring.view(M, 1, 1, 2) * scale.expand(M, L, P, 1)
            self.sampling_offsets.bias = nn.Parameter(bias.reshape(-1).clone())
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        """query (N, Lq, C); reference_points (N|1, Lq, L, 2|4) in [0,1]; input_flatten
        (N, S, C); input_spatial_shapes (L, 2) int64 (H, W); input_level_start_index (L,);
        input_padding_mask (N, S) bool, True = padding.  Returns (N, Lq, C)."""
        N, Lq, _ = query.shape
        _, S, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        if self.validate_shapes:
            assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == S

        value = _linear(self.value_proj, input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, value.shape[-1] // M)

        if _pair_core_ok(self, query, value, reference_points):
            # offsets / weights projection + softmax + locations + gather as ONE autograd node: fp32 offsets and logits
            # straight from the GEMM accumulators, read in place by the kernels (vitadapter/fused.py::_MSDAPairCore)
            out = _pair_core(self, query, value, input_spatial_shapes, input_level_start_index, reference_points)
            return _linear(self.output_proj, out)
        offsets, logits = _linear_pair(self.sampling_offsets, self.attention_weights, query)
        offsets, logits = offsets.view(N, Lq, M, L, P, 2), logits.view(N, Lq, M, L * P)
        if fused_supported(value, offsets, logits, reference_points, L, P):
            # softmax + location arithmetic + gather in one kernel (csrc/msda_fused.hip)
            out = MSDeformAttnFusedFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                  offsets, logits, reference_points)
            return _linear(self.output_proj, out)
        weights = F.softmax(logits, -1).view(N, Lq, M, L, P)

        if reference_points.shape[-1] == 2:
            wh = input_spatial_shapes.flip(-1).to(offsets.dtype)          # (L, 2) as (W, H)
            loc = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + \
                offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'
                             .format(reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                         loc, weights, self.im2col_step)
        return _linear(self.output_proj, out)
