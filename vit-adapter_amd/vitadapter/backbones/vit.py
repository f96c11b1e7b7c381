"""Plain ViT trunk used by the adapter: patch embedding, global / windowed attention blocks.

Behavioural mirror (same sub-module names => same state_dict keys, same arithmetic) of
/root/reference/detection/mmdet_custom/models/backbones/base/vit.py:39-446, which is a superset
of the segmentation copy (/root/reference/segmentation/mmseg_custom/models/backbones/base/vit.py:
39-336): det adds ``PatchEmbed(bias=)``, ``residual_indices`` / ``ResBottleneckBlock`` and the
channel-first ``LayerNorm``.  timm's ``Mlp`` / ``DropPath`` (timm 0.4.12, not in this image) are
restated here with the same attribute names.

Windowed attention keeps the reference's quirk (base/vit.py:143-156): q, k, v are projected
first and only then zero-padded up to a multiple of the window, so padded tokens take part in
the softmax with logit 0 and value 0, unmasked.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from .. import fused, kernels


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm 0.4.12 semantics: scale kept paths by 1/keep)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = float(drop_prob or 0.)

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * (mask / keep)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, drop=0.):
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(fused.linear(self.fc2, self.drop(self.act(fused.linear(self.fc1, x)))))


class PatchEmbed(nn.Module):
    """(B, C, H, W) -> ((B, H/ps * W/ps, E), H/ps, W/ps) by a stride-ps convolution."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                 norm_layer=None, flatten=True, bias=True):
        super().__init__()
        self.img_size = to_2tuple(img_size)
        self.patch_size = to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0],
                          self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size,
                              stride=self.patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        if self.flatten:
            out = fused.patch_embed(self.proj, x)          # bf16 autocast on the GPU: one GEMM on patch rows
            if out is not None:
                return self.norm(out[0]), out[1], out[2]
        x = self.proj(x)
        H, W = x.shape[-2:]
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x), H, W


class Attention(nn.Module):
    """Global multi-head self attention over all H*W tokens."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, H, W):
        B, N, C = x.shape
        qkv = fused.linear(self.qkv, x).view(B, N, 3, self.num_heads, C // self.num_heads)
        out = kernels.attention(qkv, self.scale, self.attn_drop.p if self.training else 0.)
        return self.proj_drop(fused.linear(self.proj, out.reshape(B, N, C)))


class WindowedAttention(nn.Module):
    """Self attention inside non-overlapping window_size x window_size windows."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.,
                 window_size=14, pad_mode='constant'):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.window_size = window_size
        self.pad_mode = pad_mode

    def forward(self, x, H, W):
        B, N, C = x.shape
        ws = self.window_size
        Hp, Wp = math.ceil(H / ws) * ws, math.ceil(W / ws) * ws
        nh, nw = Hp // ws, Wp // ws
        qkv = fused.linear(self.qkv, x)
        if self.pad_mode == 'constant':
            win = kernels.window_attention(qkv.view(B, N, 3, self.num_heads, C // self.num_heads),
                                           self.scale, H, W, ws, self.attn_drop.p if self.training else 0.)
            if win is not None:      # windows cut inside the kernels: no pad / partition copies
                return self.proj_drop(fused.linear(self.proj, win.reshape(B, N, C)))
        qkv = qkv.view(B, H, W, 3 * C)
        if Hp != H or Wp != W:
            if self.pad_mode == 'constant':
                qkv = F.pad(qkv, (0, 0, 0, Wp - W, 0, Hp - H))         # zeros AFTER the projection
            else:
                qkv = F.pad(qkv.permute(0, 3, 1, 2), (0, Wp - W, 0, Hp - H),
                            mode=self.pad_mode).permute(0, 2, 3, 1)
        # (B, nh, ws, nw, ws, 3C) -> (B*nh*nw, ws*ws, 3, heads, hd)
        qkv = qkv.view(B, nh, ws, nw, ws, 3 * C).permute(0, 1, 3, 2, 4, 5)
        qkv = qkv.reshape(B * nh * nw, ws * ws, 3, self.num_heads, C // self.num_heads)
        out = kernels.attention(qkv, self.scale, self.attn_drop.p if self.training else 0.)
        out = out.reshape(B, nh, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if Hp != H or Wp != W:
            out = out[:, :H, :W, :]
        return self.proj_drop(fused.linear(self.proj, out.reshape(B, N, C)))


class LayerNorm(nn.Module):
    """LayerNorm over the channel axis of (B, C, H, W) tensors (det base/vit.py:210-230)."""

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        mu = x.mean(1, keepdim=True)
        var = (x - mu).pow(2).mean(1, keepdim=True)
        x = (x - mu) / torch.sqrt(var + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class ResBottleneckBlock(nn.Module):
    """1x1 -> 3x3 -> 1x1 conv bottleneck with channel LayerNorms, last norm zero-initialised,
    no final activation (det base/vit.py:233-291)."""

    def __init__(self, in_channels, out_channels, bottleneck_channels, norm=LayerNorm,
                 act_layer=nn.GELU):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, bottleneck_channels, 1, bias=False)
        self.norm1 = norm(bottleneck_channels)
        self.act1 = act_layer()
        self.conv2 = nn.Conv2d(bottleneck_channels, bottleneck_channels, 3, padding=1, bias=False)
        self.norm2 = norm(bottleneck_channels)
        self.act2 = act_layer()
        self.conv3 = nn.Conv2d(bottleneck_channels, out_channels, 1, bias=False)
        self.norm3 = norm(out_channels)
        with torch.no_grad():
            for n in (self.norm1, self.norm2):
                n.weight.fill_(1.0)
                n.bias.zero_()
            self.norm3.weight.zero_()
            self.norm3.bias.zero_()

    def forward(self, x):
        y = self.act1(self.norm1(self.conv1(x)))
        y = self.act2(self.norm2(self.conv2(y)))
        return x + self.norm3(self.conv3(y))


class Block(nn.Module):
    """Pre-LN transformer block with optional layer scale (gamma1/gamma2), stochastic depth,
    activation checkpointing and (det) a conv residual branch."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., with_cp=False,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 windowed=False, window_size=14, use_residual=False, layer_scale=False,
                 pad_mode='constant'):
        super().__init__()
        self.with_cp = with_cp
        self.use_residual = use_residual
        self.norm1 = norm_layer(dim)
        if windowed:
            self.attn = WindowedAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias,
                                          attn_drop=attn_drop, proj_drop=drop,
                                          window_size=window_size, pad_mode=pad_mode)
        else:
            self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias,
                                  attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio),
                       act_layer=act_layer, drop=drop)
        self.layer_scale = layer_scale
        if layer_scale:
            self.gamma1 = nn.Parameter(torch.ones(dim), requires_grad=True)
            self.gamma2 = nn.Parameter(torch.ones(dim), requires_grad=True)
        if use_residual:
            self.residual = ResBottleneckBlock(in_channels=dim, out_channels=dim,
                                               bottleneck_channels=dim // 2, norm=LayerNorm,
                                               act_layer=act_layer)

    def _body(self, x, H, W, h=None, next_norm=None):
        # x + drop_path(gamma1 * attn(norm1(x))), x + drop_path(gamma2 * mlp(norm2(x)))
        # (ref base/vit.py:301-306); fused.* fall back to exactly that expression off the bf16 path.
        # ``h`` = norm1(x) when the caller already has it; with ``next_norm`` the result is
        # (x, next_norm(x)): the residual update and the next LayerNorm run as one kernel.
        g1, g2 = (self.gamma1, self.gamma2) if self.layer_scale else (None, None)
        if h is None:
            x, h = fused.layer_norm_keep(self.norm1, x)
        x, h = fused.residual_ln(x, self.attn(h, H, W), g1, self.drop_path, self.norm2)
        f = self.mlp(h)
        if next_norm is not None and not self.use_residual:
            return fused.residual_ln(x, f, g2, self.drop_path, next_norm)
        x = fused.residual(x, f, g2, self.drop_path)
        if self.use_residual:
            B, N, C = x.shape
            y = self.residual(x.reshape(B, H, W, C).permute(0, 3, 1, 2))
            x = y.permute(0, 2, 3, 1).reshape(B, N, C)
        return x if next_norm is None else fused.layer_norm_keep(next_norm, x)

    def forward(self, x, H, W):
        if self.with_cp and x.requires_grad:
            return cp.checkpoint(self._body, x, H, W, use_reentrant=False)
        return self._body(x, H, W)

    def forward_chain(self, x, H, W, h=None, next_norm=None):
        """``forward`` for a run of blocks: takes norm1(x) from the previous block (``h``) and
        returns (x, next_norm(x)) for the next one (``next_norm`` = its norm1), or x at the end."""
        if self.with_cp and x.requires_grad:
            x = cp.checkpoint(self._body, x, H, W, use_reentrant=False)
            return x if next_norm is None else fused.layer_norm_keep(next_norm, x)
        return self._body(x, H, W, h, next_norm)


def run_blocks(blocks, x, H, W, h=None):
    """x through ``blocks`` with each residual update fused to the LayerNorm that follows it."""
    blocks = list(blocks)
    for i, blk in enumerate(blocks):
        nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else None
        out = blk.forward_chain(x, H, W, h, nxt)
        x, h = out if nxt is not None else (out, None)
    return x


class TIMMVisionTransformer(nn.Module):
    """ViT trunk (no cls token, no head): patch_embed, pos_embed (1, 1 + num_patches, E), blocks."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, residual_indices=(),
                 embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., layer_scale=True,
                 embed_layer=PatchEmbed, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 act_layer=nn.GELU, window_attn=False, window_size=14, with_cp=False,
                 pretrained=None, pad_mode='constant'):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.norm_layer = norm_layer
        self.act_layer = act_layer
        self.pretrain_size = img_size
        self.drop_path_rate = drop_path_rate
        self.drop_rate = drop_rate

        window_attn = list(window_attn) if isinstance(window_attn, (list, tuple)) else [window_attn] * depth
        window_size = list(window_size) if isinstance(window_size, (list, tuple)) else [window_size] * depth

        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size,
                                       in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)

        dpr = torch.linspace(0, drop_path_rate, depth).tolist()
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i],
                  norm_layer=norm_layer, act_layer=act_layer, windowed=window_attn[i],
                  window_size=window_size[i], layer_scale=layer_scale, with_cp=with_cp,
                  use_residual=i in residual_indices, pad_mode=pad_mode)
            for i in range(depth)])
        self.init_weights(pretrained)

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, map_location='cpu', strict=False)
