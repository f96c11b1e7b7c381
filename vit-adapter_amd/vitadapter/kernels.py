"""Op dispatch of the backbone's hot operators onto libvitadapter_hip.so.

Every function here takes and returns torch tensors and enqueues work on torch's current
stream.  Importing this module loads the HIP library (``_vah``); if it is missing the import
fails -- there is no CPU fallback behind these entry points.
"""
import torch

import _vah  # noqa: F401  (hard requirement: raises ImportError when the .so is not built)


def _stream(t):
    return _vah.raw_stream(t.device)


def _attention_math(qkv, scale, dropout_p=0.):
    """Library-GEMM statement of the same arithmetic (fp32 models, head_dim != 64, dropout)."""
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)          # each (B, heads, N, hd)
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    if dropout_p > 0.:
        attn = torch.nn.functional.dropout(attn, dropout_p, True)
    return (attn @ v).transpose(1, 2)


class _FlashAttention(torch.autograd.Function):
    """bf16 MFMA attention on the packed qkv projection (csrc/attn_fwd.hip, attn_bwd.hip)."""

    @staticmethod
    def forward(ctx, qkv, scale):
        B, N, three, H, hd = qkv.shape
        qkv = qkv.contiguous()
        C = H * hd
        out = torch.empty((B, N, H, hd), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        Np = _vah.lib.vah_attn_padded_len(N)
        ws = torch.empty((B * H * hd * Np,), dtype=qkv.dtype, device=qkv.device)
        base = qkv.data_ptr()
        esz = qkv.element_size()
        with _vah.on(qkv.device):
            rc = _vah.lib.vah_attn_fwd_bf16(base, base + C * esz, base + 2 * C * esz, 3 * C, N * 3 * C,
                                            B, H, N, float(scale), ws.data_ptr(), out.data_ptr(), C,
                                            lse.data_ptr(), _stream(qkv))
        _vah.check(rc, 'vah_attn_fwd_bf16')
        ctx.save_for_backward(qkv, out, lse)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        return _attention_backward(qkv, out, lse, dout.contiguous().to(qkv.dtype), ctx.scale), None


def _attention_backward(qkv, out, lse, dout, scale):
    """d(loss)/d(qkv) as one packed (B, N, 3, heads, 64) tensor (csrc/attn_bwd.hip)."""
    B, N, _, H, hd = qkv.shape
    C = H * hd
    dqkv = torch.empty_like(qkv)
    ws = torch.empty((_vah.lib.vah_attn_bwd_workspace_bytes(B, H, N),), dtype=torch.uint8,
                     device=qkv.device)
    base, dbase, esz = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
    with _vah.on(qkv.device):
        rc = _vah.lib.vah_attn_bwd_bf16(
            base, base + C * esz, base + 2 * C * esz, 3 * C, N * 3 * C, out.data_ptr(),
            dout.data_ptr(), C, lse.data_ptr(), B, H, N, float(scale), ws.data_ptr(),
            dbase, dbase + C * esz, dbase + 2 * C * esz, 3 * C, N * 3 * C, _stream(qkv))
    _vah.check(rc, 'vah_attn_bwd_bf16')
    return dqkv


class _FlashAttentionBias(torch.autograd.Function):
    """softmax(q k^T * scale + bias) v with a (heads, N, N) bias shared by the batch - BEiT's relative position bias
    (base/beit.py:120-144) - on the MFMA kernels of csrc/attn_flash.hip.  The bias enters the kernels as bf16 times
    log2(e), padded to 64-column rows, plus its transpose for the dK / dV pass; its gradient leaves the dQ pass as dS per
    image (bf16) and is summed over the batch here."""

    @staticmethod
    def forward(ctx, qkv, bias, scale):
        B, N, three, H, hd = qkv.shape
        qkv = qkv.contiguous()
        C = H * hd
        Np = (N + 63) // 64 * 64
        b2 = bias.detach().float() * 1.4426950408889634
        bl = torch.zeros((H, N, Np), dtype=torch.bfloat16, device=qkv.device)
        bl[:, :, :N] = b2
        blt = torch.zeros((H, N, Np), dtype=torch.bfloat16, device=qkv.device)
        blt[:, :, :N] = b2.transpose(1, 2)
        out = torch.empty((B, N, H, hd), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        base, esz = qkv.data_ptr(), qkv.element_size()
        with _vah.on(qkv.device):
            rc = _vah.lib.vah_attn_bias_fwd_bf16(base, base + C * esz, base + 2 * C * esz, 3 * C, N * 3 * C, B, H, N, float(scale),
                                                 bl.data_ptr(), Np, out.data_ptr(), C, lse.data_ptr(), _stream(qkv))
        _vah.check(rc, 'vah_attn_bias_fwd_bf16')
        ctx.save_for_backward(qkv, out, lse, bl, blt)
        ctx.scale = float(scale)
        ctx.bias_dtype = bias.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, bl, blt = ctx.saved_tensors
        B, N, _, H, hd = qkv.shape
        C, Np = H * hd, bl.shape[-1]
        dout = dout.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        ds = torch.empty((B, H, N, Np), dtype=torch.bfloat16, device=qkv.device)
        delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        base, dbase, esz = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
        with _vah.on(qkv.device):
            rc = _vah.lib.vah_attn_bias_bwd_bf16(
                base, base + C * esz, base + 2 * C * esz, 3 * C, N * 3 * C, out.data_ptr(), dout.data_ptr(), C, lse.data_ptr(),
                B, H, N, ctx.scale, bl.data_ptr(), blt.data_ptr(), Np, ds.data_ptr(), delta.data_ptr(), dbase, dbase + C * esz,
                dbase + 2 * C * esz, 3 * C, N * 3 * C, _stream(qkv))
        _vah.check(rc, 'vah_attn_bias_bwd_bf16')
        dbias = None
        if ctx.needs_input_grad[1]:
            dbias = ds[..., :N].float().sum(0).to(ctx.bias_dtype)
        return dqkv, dbias, None


class _FlashAttentionRelPos(torch.autograd.Function):
    """_FlashAttentionBias with the bias given as BEiT stores it: a (T, heads) table and an (N, N) index
    (base/beit.py:120-131).  The two bf16 operands of the kernels are built straight from the table and the table's
    gradient is reduced straight from the dQ pass's dS (csrc/relpos.hip): no (heads, N, N) fp32 tensor either way."""

    @staticmethod
    def forward(ctx, qkv, table, index, scale):
        B, N, three, H, hd = qkv.shape
        qkv = qkv.contiguous()
        C = H * hd
        Np = (N + 63) // 64 * 64
        tb = table.detach().float().contiguous()
        index = index.contiguous()
        bl = torch.empty((H, N, Np), dtype=torch.bfloat16, device=qkv.device)
        blt = torch.empty((H, N, Np), dtype=torch.bfloat16, device=qkv.device)
        out = torch.empty((B, N, H, hd), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        base, esz = qkv.data_ptr(), qkv.element_size()
        with _vah.on(qkv.device):
            _vah.check(_vah.lib.vah_relpos_bias_build(tb.data_ptr(), index.data_ptr(), tb.shape[0], H, N, Np, bl.data_ptr(),
                                                      blt.data_ptr(), _stream(qkv)), 'vah_relpos_bias_build')
            rc = _vah.lib.vah_attn_bias_fwd_bf16(base, base + C * esz, base + 2 * C * esz, 3 * C, N * 3 * C, B, H, N, float(scale),
                                                 bl.data_ptr(), Np, out.data_ptr(), C, lse.data_ptr(), _stream(qkv))
        _vah.check(rc, 'vah_attn_bias_fwd_bf16')
        ctx.save_for_backward(qkv, out, lse, bl, blt, index)
        ctx.scale = float(scale)
        ctx.table_meta = (table.shape[0], table.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, bl, blt, index = ctx.saved_tensors
        B, N, _, H, hd = qkv.shape
        C, Np = H * hd, bl.shape[-1]
        T, tdtype = ctx.table_meta
        dout = dout.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        ds = torch.empty((B, H, N, Np), dtype=torch.bfloat16, device=qkv.device)
        delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        base, dbase, esz = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
        dtable = None
        with _vah.on(qkv.device):
            rc = _vah.lib.vah_attn_bias_bwd_bf16(
                base, base + C * esz, base + 2 * C * esz, 3 * C, N * 3 * C, out.data_ptr(), dout.data_ptr(), C, lse.data_ptr(),
                B, H, N, ctx.scale, bl.data_ptr(), blt.data_ptr(), Np, ds.data_ptr(), delta.data_ptr(), dbase, dbase + C * esz,
                dbase + 2 * C * esz, 3 * C, N * 3 * C, _stream(qkv))
            _vah.check(rc, 'vah_attn_bias_bwd_bf16')
            if ctx.needs_input_grad[1]:
                dtable = torch.empty((T, H), dtype=torch.float32, device=qkv.device)
                ws = torch.empty((_vah.lib.vah_relpos_bias_grad_ws_floats(T, H),), dtype=torch.float32, device=qkv.device)
                _vah.check(_vah.lib.vah_relpos_bias_grad(ds.data_ptr(), index.data_ptr(), B, H, N, Np, T, ws.data_ptr(),
                                                         dtable.data_ptr(), _stream(qkv)), 'vah_relpos_bias_grad')
                dtable = dtable.to(tdtype)
        return dqkv, dtable, None, None


_RESIDENT_WINDOW = 224    # tokens per window served by the one-kernel resident path (include/vitadapter_hip.h)


class _WindowFlashAttention(torch.autograd.Function):
    """bf16 MFMA attention inside win x win windows of a (B, gh, gw) token grid, windows cut by the
    kernels' addressing (no pad / partition / merge / crop copies)."""

    @staticmethod
    def forward(ctx, qkv, scale, gh, gw, win):
        B, N, three, H, hd = qkv.shape
        qkv = qkv.contiguous()
        C = H * hd
        Z = B * (-(-gh // win)) * (-(-gw // win))
        Nw = win * win
        out = torch.empty((B, N, H, hd), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((Z, H, Nw), dtype=torch.float32, device=qkv.device)
        # windows of <= 224 tokens stay resident in LDS (csrc/attn_win.hip): no V^T workspace
        ws = (torch.empty((Z * H * hd * _vah.lib.vah_attn_padded_len(Nw),), dtype=qkv.dtype, device=qkv.device)
              if Nw > _RESIDENT_WINDOW else None)
        base, esz = qkv.data_ptr(), qkv.element_size()
        with _vah.on(qkv.device):
            rc = _vah.lib.vah_attn_win_fwd_bf16(base, base + C * esz, base + 2 * C * esz, 3 * C, B, gh, gw,
                                                win, H, float(scale), ws.data_ptr() if ws is not None else 0, out.data_ptr(), C,
                                                lse.data_ptr(), _stream(qkv))
        _vah.check(rc, 'vah_attn_win_fwd_bf16')
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (float(scale), gh, gw, win, Z, Nw)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        scale, gh, gw, win, Z, Nw = ctx.cfg
        B, N, _, H, hd = qkv.shape
        C = H * hd
        dout = dout.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        ws = (torch.empty((_vah.lib.vah_attn_bwd_workspace_bytes(Z, H, Nw),), dtype=torch.uint8, device=qkv.device)
              if Nw > _RESIDENT_WINDOW else None)
        base, dbase, esz = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
        with _vah.on(qkv.device):
            rc = _vah.lib.vah_attn_win_bwd_bf16(
                base, base + C * esz, base + 2 * C * esz, 3 * C, out.data_ptr(), dout.data_ptr(), C,
                lse.data_ptr(), B, gh, gw, win, H, scale, ws.data_ptr() if ws is not None else 0, dbase, dbase + C * esz,
                dbase + 2 * C * esz, 3 * C, _stream(qkv))
        _vah.check(rc, 'vah_attn_win_bwd_bf16')
        return dqkv, None, None, None, None


def window_attention(qkv, scale, gh, gw, win, dropout_p=0.):
    """Windowed attention on the packed projection of a (B, gh*gw) token grid; returns
    (B, gh*gw, heads, head_dim) or None when the fused path does not apply (the caller then uses
    the reference's pad / partition sequence)."""
    if (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 64 and dropout_p == 0.
            and qkv.shape[1] == gh * gw and qkv.numel() > 0 and 1 <= win <= 64
            and not FLAGS['force_math_attention'] and FLAGS['fused_windows']):
        return _WindowFlashAttention.apply(qkv, scale, gh, gw, win)
    return None


def attention_bias(qkv, bias, scale, dropout_p=0.):
    """softmax(q k^T * scale + bias) v on a packed projection (B, N, 3, heads, head_dim), bias (heads, N, N) shared by
    the batch; returns (B, N, heads, head_dim), or None when the MFMA path does not apply (the caller then evaluates the
    reference expression)."""
    if (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 64 and dropout_p == 0. and qkv.shape[1] > 0
            and bias is not None and tuple(bias.shape) == (qkv.shape[3], qkv.shape[1], qkv.shape[1])
            and not FLAGS['force_math_attention']):
        return _FlashAttentionBias.apply(qkv, bias, scale)
    return None


def attention_relpos(qkv, table, index, scale, dropout_p=0.):
    """attention_bias with bias[h][i][j] = table[index[i][j]][h] (BEiT); None when the MFMA path does not apply."""
    N, H = qkv.shape[1], qkv.shape[3]
    if (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 64 and dropout_p == 0. and N > 0 and table.dim() == 2
            and table.shape[1] == H and index.dtype == torch.int64 and tuple(index.shape) == (N, N)
            and table.shape[0] * 4 <= 150 * 1024 and not FLAGS['force_math_attention']):
        return _FlashAttentionRelPos.apply(qkv, table, index, scale)
    return None


def attention(qkv, scale, dropout_p=0.):
    """Softmax attention on a packed projection.

    qkv: (B, N, 3, heads, head_dim) view of the fused qkv Linear output (any strides).
    Returns (B, N, heads, head_dim).  Arithmetic of the reference's Attention / WindowedAttention
    (/root/reference/detection/mmdet_custom/models/backbones/base/vit.py:83-88,154-159):
    softmax(q k^T * scale) v, with dropout on the probabilities in training.
    """
    if (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 64 and dropout_p == 0.
            and qkv.shape[1] > 0 and not FLAGS['force_math_attention']):
        return _FlashAttention.apply(qkv, scale)
    return _attention_math(qkv, scale, dropout_p)


FLAGS = {'force_math_attention': False, 'fused_windows': True}
