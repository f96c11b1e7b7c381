"""SpatialPriorModule on NHWC bf16 activations: implicit-GEMM 3x3 convolutions (csrc/conv.hip) with the
BatchNorm + ReLU, max-pool and image-layout kernels of csrc/spm_nhwc.hip between them.

Reference: SpatialPriorModule.forward of
/root/reference/detection/mmdet_custom/models/backbones/adapter_modules.py:217-268.  The arithmetic is the module's
(convolutions with bf16 operands and fp32 accumulation as under autocast, batch statistics in fp32, SyncBatchNorm
all-reduces of the sums); the layout is the kernels': no NCHW <-> NHWC conversion inside the module, and the
stride-8/16/32 maps are already the (B, H*W, C) token rows the 1x1 projections and the adapter consume.
There is no CPU path behind these functions."""
import torch
import torch.nn.functional as F

import _vah
from . import conv, fused


def _stream(t):
    return _vah.raw_stream(t.device)


def image_to_nhwc16(x):
    """(N, 3, H, W) fp32 -> (N, H, W, 16) bf16 with channels 3..15 zero (no gradient: the image is a leaf input)."""
    N, C, H, W = x.shape
    assert C == 3 and x.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty((N, H, W, 16), dtype=torch.bfloat16, device=x.device)
    with _vah.on(x.device):
        _vah.check(_vah.lib.vah_image_to_nhwc16_bf16(x.data_ptr(), N, H, W, y.data_ptr(), _stream(x)), 'image_to_nhwc16')
    return y


class _Conv3x3(torch.autograd.Function):
    """nn.Conv2d(k=3, padding=1, bias=False) on NHWC bf16; weight (Cout, Cin, 3, 3) fp32.  An input with more channels
    than the weight (the 16-channel image) is matched by zero weight columns."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        cin = x.shape[-1]
        w = weight.detach()
        if w.shape[1] != cin:
            w = F.pad(w, (0, 0, 0, 0, 0, cin - w.shape[1]))
        wb = w.to(torch.bfloat16)
        ctx.save_for_backward(x, wb)
        ctx.stride, ctx.wcin = stride, weight.shape[1]
        return conv.conv3x3_forward(x, conv.forward_weight(wb), stride)

    @staticmethod
    def backward(ctx, gy):
        x, wb = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = conv.conv3x3_input_grad(gy, conv.dgrad_weight(wb), ctx.stride, x.shape[1:3])
        if ctx.needs_input_grad[1]:
            gw = conv.conv3x3_weight_grad(x, gy, ctx.stride).permute(0, 3, 1, 2)[:, :ctx.wcin].contiguous()
        return gx, gw, None


class _BNRelu(torch.autograd.Function):
    """relu(BatchNorm(x)) over the rows of an NHWC tensor; the statistics / SyncBatchNorm protocol of fused._BNTail."""

    @staticmethod
    def forward(ctx, x, weight, bias, norm, relu):
        C = x.shape[-1]
        rows = x.numel() // C
        x = x.contiguous()
        dev, st = x.device, _stream(x)
        training = norm.training or norm.running_mean is None
        group = fused._sync_group(norm) if training else None
        w = weight.detach().float().contiguous() if weight is not None else None
        b = bias.detach().float().contiguous() if bias is not None else None
        with _vah.on(dev):
            if training:
                sums = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
                ws = torch.empty(_vah.lib.vah_bn_nhwc_ws_floats(C), dtype=torch.float32, device=dev)
                _vah.check(_vah.lib.vah_bn_nhwc_stats(x.data_ptr(), rows, C, sums.data_ptr(), ws.data_ptr(), st), 'bn_nhwc_stats')
                sums[2 * C:].fill_(float(rows))
                if group is not None:
                    import torch.distributed as dist
                    dist.all_reduce(sums, group=group)
                mean = torch.empty(C, dtype=torch.float32, device=dev)
                rstd = torch.empty(C, dtype=torch.float32, device=dev)
                track = norm.running_mean is not None
                _vah.check(_vah.lib.vah_bn_finalize_stats(
                    sums.data_ptr(), C, float(norm.eps), float(norm.momentum),
                    norm.running_mean.data_ptr() if track else None, norm.running_var.data_ptr() if track else None,
                    mean.data_ptr(), rstd.data_ptr(), st), 'bn_finalize_stats')
                if track:
                    with torch.no_grad():
                        norm.num_batches_tracked += 1
                count = sums[2 * C:]
            else:
                count = None
                mean = norm.running_mean.float().contiguous()
                rstd = torch.rsqrt(norm.running_var.float() + norm.eps)
            y = torch.empty_like(x)
            _vah.check(_vah.lib.vah_bn_nhwc_apply(x.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(),
                                                  w.data_ptr() if w is not None else None,
                                                  b.data_ptr() if b is not None else None, int(relu), y.data_ptr(), st),
                       'bn_nhwc_apply')
        ctx.save_for_backward(x, mean, rstd, w, b, count)
        ctx.meta = (training, group, weight is not None, bias is not None, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w, b, count = ctx.saved_tensors
        training, group, has_w, has_b, relu = ctx.meta
        C = x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous().to(torch.bfloat16)
        dev, st = x.device, _stream(x)
        wp = w.data_ptr() if w is not None else None
        bp = b.data_ptr() if b is not None else None
        with _vah.on(dev):
            sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
            ws = torch.empty(_vah.lib.vah_bn_nhwc_ws_floats(C), dtype=torch.float32, device=dev)
            _vah.check(_vah.lib.vah_bn_nhwc_bwd_stats(x.data_ptr(), dy.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(), wp,
                                                      bp, int(relu), sums.data_ptr(), ws.data_ptr(), st), 'bn_nhwc_bwd_stats')
            local = sums.clone() if (training and group is not None) else sums      # dweight / dbias are per-rank sums
            dweight = local[C:] if has_w else None
            dbias = local[:C] if has_b else None
            dx = None
            if ctx.needs_input_grad[0]:
                if training:
                    if group is not None:
                        import torch.distributed as dist
                        dist.all_reduce(sums, group=group)
                    means = sums / count
                else:
                    means = torch.zeros_like(sums)          # running statistics are constants
                dx = torch.empty_like(x)
                _vah.check(_vah.lib.vah_bn_nhwc_bwd_apply(x.data_ptr(), dy.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(),
                                                          wp, bp, int(relu), means[:C].data_ptr(), means[C:].data_ptr(),
                                                          dx.data_ptr(), st), 'bn_nhwc_bwd_apply')
        return dx, dweight, dbias, None, None


class _MaxPool(torch.autograd.Function):
    """MaxPool2d(3, stride 2, padding 1) on NHWC bf16."""

    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        x = x.contiguous()
        y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
        idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        with _vah.on(x.device):
            _vah.check(_vah.lib.vah_maxpool3s2_nhwc_fwd_bf16(x.data_ptr(), N, H, W, C, y.data_ptr(), idx.data_ptr(), _stream(x)),
                       'maxpool_nhwc_fwd')
        ctx.save_for_backward(idx)
        ctx.in_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        idx, = ctx.saved_tensors
        N, H, W, C = ctx.in_shape
        gy = gy.contiguous().to(torch.bfloat16)
        gx = torch.empty(ctx.in_shape, dtype=torch.bfloat16, device=gy.device)
        with _vah.on(gy.device):
            _vah.check(_vah.lib.vah_maxpool3s2_nhwc_bwd_bf16(gy.data_ptr(), idx.data_ptr(), N, H, W, C, gx.data_ptr(), _stream(gy)),
                       'maxpool_nhwc_bwd')
        return gx


class _Conv1x1ToPlanes(torch.autograd.Function):
    """1x1 convolution without bias from an NHWC bf16 map to NCHW bf16 planes as plain GEMMs:
    out[b] (Co x HW) = W (Co x Ci) x[b]^T, x[b] the (HW x Ci) token rows."""

    @staticmethod
    def forward(ctx, x, weight):
        B, H, W, Ci = x.shape
        Co = weight.shape[0]
        wb = fused.BF16_COPIES.get(weight).view(Co, Ci)
        out = torch.empty((B, Co, H, W), dtype=torch.bfloat16, device=x.device)
        for b in range(B):
            fused.gemm_bf16(wb, x[b].view(H * W, Ci), trans_b=True, out=out[b].view(Co, H * W))
        ctx.save_for_backward(x, wb)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wb = ctx.saved_tensors
        B, H, W, Ci = x.shape
        Co = wb.shape[0]
        g = g.contiguous().to(torch.bfloat16)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            for b in range(B):
                fused.gemm_bf16(g[b].view(Co, H * W), wb, trans_a=True, out=dx[b].view(H * W, Ci))
        if ctx.needs_input_grad[1]:
            for b in range(B):
                part = fused.gemm_bf16(g[b].view(Co, H * W), x[b].view(H * W, Ci), out_dtype=torch.float32)
                dw = part if dw is None else dw.add_(part)
            dw = dw.view(Co, Ci, 1, 1)
        return dx, dw


def usable(spm, x):
    """The NHWC path serves the module as the reference builds it (3x3 / padding 1 / bias-free convolutions with
    64-multiple widths, (Sync)BatchNorm + ReLU, the 3/2/1 max-pool) on a CUDA fp32 image under bf16 autocast."""
    # the image's own gradient is not produced by this path: an input that requires grad takes the module as written
    if not (fused.ENABLED.get('spm_nhwc', True) and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and x.dim() == 4 and x.shape[1] == 3
            and fused._bf16_autocast() and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0 and x.numel() > 0):
        return False
    convs = [spm.stem[0], spm.stem[3], spm.stem[6], spm.conv2[0], spm.conv3[0], spm.conv4[0]]
    for i, c in enumerate(convs):
        if not (isinstance(c, torch.nn.Conv2d) and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.bias is None
                and c.groups == 1 and c.dilation == (1, 1) and c.out_channels % 64 == 0 and c.out_channels <= 256
                and (c.out_channels & (c.out_channels - 1)) == 0 and (i == 0 or c.in_channels % 64 == 0)
                and c.weight.dtype == torch.float32):
            return False
    norms = [spm.stem[1], spm.stem[4], spm.stem[7], spm.conv2[1], spm.conv3[1], spm.conv4[1]]
    return all(isinstance(n, torch.nn.modules.batchnorm._BatchNorm) and n.momentum is not None
               and (n.training or n.running_mean is not None) for n in norms)


def _cbr(conv_mod, norm, x):
    y = _Conv3x3.apply(x, conv_mod.weight, conv_mod.stride[0])
    return _BNRelu.apply(y, norm.weight, norm.bias, norm, True)


def forward(spm, x, level_embed):
    """-> (c1, c): c1 = fc1's output WITHOUT its bias as NCHW bf16 planes (the caller folds the bias into the
    BatchNorm tail), c = cat([fc_l(c_l) + level_embed[l-2] for l = 2, 3, 4]) as (B, T, E) fp32 token rows."""
    t = image_to_nhwc16(x)
    t = _cbr(spm.stem[0], spm.stem[1], t)
    t = _cbr(spm.stem[3], spm.stem[4], t)
    t = _cbr(spm.stem[6], spm.stem[7], t)
    c1 = _MaxPool.apply(t)
    c2 = _cbr(spm.conv2[0], spm.conv2[1], c1)
    c3 = _cbr(spm.conv3[0], spm.conv3[1], c2)
    c4 = _cbr(spm.conv4[0], spm.conv4[1], c3)
    B = x.shape[0]
    toks = []
    for l, (fc, c) in enumerate(((spm.fc2, c2), (spm.fc3, c3), (spm.fc4, c4))):
        w = fc.weight.view(fc.weight.shape[0], fc.weight.shape[1])
        toks.append(fused._LinearBF16.apply(c.view(B, -1, c.shape[-1]), w, fc.bias + level_embed[l]))
    return _Conv1x1ToPlanes.apply(c1, spm.fc1.weight), torch.cat(toks, dim=1).float()
