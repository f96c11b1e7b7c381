"""Fused memory-bound operators of the blocks (csrc/fused_ops.hip) with their autograd glue.

Each helper computes exactly what the reference's module code computes (cited at the call sites);
it takes the fused HIP path when the tensors are what bf16 autocast produces on the GPU (fp32
residual stream, bf16 branch outputs) and otherwise evaluates the same expression with torch ops
(fp32 runs, CPU host-logic tests).
"""
import torch
import torch.nn.functional as F

import _vah

ENABLED = {'layer_norm': True, 'residual': True, 'dwconv': True}


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _scratch(K, device):
    """Partial-sum scratch for the column reductions (stream-ordered: allocated per call from
    torch's caching allocator, so concurrent streams never share it)."""
    return torch.empty(_vah.lib.vah_reduce_ws_floats(K), dtype=torch.float32, device=device)


def _bf16_autocast():
    return torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16


class _LayerNormBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[-1]
        x2 = x.contiguous().view(-1, C)
        rows = x2.shape[0]
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        w, b = weight.contiguous(), bias.contiguous()
        with torch.cuda.device(x.device):
            _vah.check(_vah.lib.vah_layernorm_fwd_f32_bf16(
                x2.data_ptr(), w.data_ptr(), b.data_ptr(), rows, C, float(eps), y.data_ptr(),
                mean.data_ptr(), rstd.data_ptr(), _stream(x)), 'layernorm_fwd')
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, g):
        x2, w, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        g = g.contiguous().to(torch.bfloat16)
        dx = torch.empty_like(x2)
        dwb = torch.empty(2, C, dtype=torch.float32, device=x2.device)
        ws = _scratch(2 * C, x2.device)
        with torch.cuda.device(x2.device):
            _vah.check(_vah.lib.vah_layernorm_bwd_f32_bf16(
                x2.data_ptr(), g.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C,
                dx.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr(), ws.data_ptr(), _stream(x2)),
                'layernorm_bwd')
        return dx.view(ctx.shape), dwb[0], dwb[1], None


def layer_norm(norm, x):
    """``norm(x)`` for an nn.LayerNorm; bf16 output when the consumer is a bf16 GEMM (autocast)."""
    if (ENABLED['layer_norm'] and x.is_cuda and x.dtype == torch.float32 and _bf16_autocast()
            and isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048 and x.numel() > 0):
        return _LayerNormBF16.apply(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


class _ScaleResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, z, gamma, s):
        B, C = x.shape[0], x.shape[-1]
        rpb = x.numel() // (B * C)
        x, z = x.contiguous(), z.contiguous()
        y = torch.empty_like(x)
        gp = gamma.contiguous() if gamma is not None else None
        with torch.cuda.device(x.device):
            _vah.check(_vah.lib.vah_scale_residual_fwd(
                x.data_ptr(), z.data_ptr(), gp.data_ptr() if gp is not None else None,
                s.data_ptr() if s is not None else None, B, rpb, C, y.data_ptr(), _stream(x)),
                'scale_residual_fwd')
        ctx.save_for_backward(z, gp, s)
        ctx.dims = (B, rpb, C)
        return y

    @staticmethod
    def backward(ctx, g):
        z, gp, s = ctx.saved_tensors
        B, rpb, C = ctx.dims
        g = g.contiguous()
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=g.device) if gp is not None else None
        ws = _scratch(C, g.device) if gp is not None else None
        with torch.cuda.device(g.device):
            _vah.check(_vah.lib.vah_scale_residual_bwd(
                g.data_ptr(), z.data_ptr(), gp.data_ptr() if gp is not None else None,
                s.data_ptr() if s is not None else None, B, rpb, C, dz.data_ptr(),
                dgamma.data_ptr() if dgamma is not None else None,
                ws.data_ptr() if ws is not None else None, _stream(g)), 'scale_residual_bwd')
        return g, dz, dgamma, None


def residual(x, z, gamma=None, drop_path=None):
    """``x + drop_path(gamma * z)`` (gamma / drop_path optional), the residual update of the
    reference's Block / Injector / Extractor."""
    prob = float(getattr(drop_path, 'drop_prob', 0.) or 0.)
    training = bool(getattr(drop_path, 'training', False))
    if (ENABLED['residual'] and x.is_cuda and x.dtype == torch.float32 and z.dtype == torch.bfloat16
            and x.shape == z.shape and x.shape[-1] % 4 == 0 and x.numel() > 0
            and (gamma is None or gamma.dtype == torch.float32)):
        s = None
        if prob > 0. and training:
            keep = 1.0 - prob
            s = x.new_empty((x.shape[0],)).bernoulli_(keep).div_(keep)
        return _ScaleResidual.apply(x, z, gamma, s)
    t = gamma * z if gamma is not None else z
    return x + (drop_path(t) if drop_path is not None else t)


class _DWConvTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, H, W):
        B, N, C = x.shape
        x = x.contiguous()
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous() if bias is not None else None
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _vah.check(_vah.lib.vah_dwconv3x3_tokens_bf16(
                x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, B, H, W, C, 0,
                y.data_ptr(), _stream(x)), 'dwconv_fwd')
        ctx.save_for_backward(x, w)
        ctx.dims = (B, H, W, C, bias is not None, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, H, W, C, has_bias, wdtype = ctx.dims
        g = g.contiguous().to(torch.bfloat16)
        dx = torch.empty_like(x)
        dw = torch.empty(C * 9 + C, dtype=torch.float32, device=x.device)
        ws = _scratch(10 * C, x.device)
        with torch.cuda.device(x.device):
            _vah.check(_vah.lib.vah_dwconv3x3_tokens_bf16(
                g.data_ptr(), w.data_ptr(), None, B, H, W, C, 1, dx.data_ptr(), _stream(x)), 'dwconv_dgrad')
            _vah.check(_vah.lib.vah_dwconv3x3_tokens_wgrad_bf16(
                x.data_ptr(), g.data_ptr(), B, H, W, C, dw.data_ptr(),
                dw[C * 9:].data_ptr() if has_bias else None, ws.data_ptr(), _stream(x)), 'dwconv_wgrad')
        return (dx, dw[:C * 9].view(C, 1, 3, 3).to(wdtype),
                dw[C * 9:].to(wdtype) if has_bias else None, None, None)


def dwconv_tokens(conv, x, H, W):
    """ConvFFN's DWConv on the concatenated token maps; None when the fused path does not apply."""
    B, N, C = x.shape
    if (ENABLED['dwconv'] and x.is_cuda and x.dtype == torch.bfloat16 and C % 4 == 0 and C <= 1024
            and H % 2 == 0 and W % 2 == 0 and N == 21 * (H // 2) * (W // 2) and x.numel() > 0
            and conv.weight.shape == (C, 1, 3, 3)):
        return _DWConvTokens.apply(x, conv.weight, conv.bias, H, W)
    return None
