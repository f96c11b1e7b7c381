"""BEiT trunk for the adapter.

Interface mirror of /root/reference/segmentation/mmseg_custom/models/backbones/base/beit.py
(Attention :61-152, Block :155-191, RelativePositionBias :259-292, BEiT :296-378): same
constructor arguments, same parameter / buffer names (``attn.qkv.weight`` without bias,
``attn.q_bias`` / ``attn.v_bias``, ``attn.relative_position_bias_table`` +
``relative_position_index``, ``gamma_1`` / ``gamma_2``, ``cls_token``), so reference checkpoints
load key for key.

Attention here is  softmax(q k^T * scale + B) v  with a learned relative-position bias B of shape
(heads, N, N) - the token count is fixed by ``img_size`` (beit.py:80-106: the bias table is built
for the patch grid of ``img_size`` plus the class token).  It runs through torch's
scaled_dot_product_attention with the bias as an additive mask on the GPU (the flash kernels of
csrc/attn_*.hip take no bias yet); LayerNorm, residual updates, Linear layers and the whole adapter
side use the same HIP paths as ViTAdapter.
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from .. import fused, kernels
from .vit import DropPath, to_2tuple


def relative_position_index(window_size):
    """(Wh*Ww + 1)^2 index table into the bias table: pairwise offsets of the patch grid plus three
    extra entries for cls->token, token->cls and cls->cls (beit.py:86-101)."""
    Wh, Ww = window_size
    n_rel = (2 * Wh - 1) * (2 * Ww - 1) + 3
    ys, xs = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing='ij')
    coords = torch.stack([ys.flatten(), xs.flatten()])               # (2, Wh*Ww)
    rel = coords[:, :, None] - coords[:, None, :]                      # (2, n, n)
    idx = (rel[0] + Wh - 1) * (2 * Ww - 1) + (rel[1] + Ww - 1)
    out = torch.zeros((Wh * Ww + 1,) * 2, dtype=idx.dtype)
    out[1:, 1:] = idx
    out[0, :] = n_rel - 3
    out[:, 0] = n_rel - 2
    out[0, 0] = n_rel - 1
    return out, n_rel


class Mlp(nn.Module):
    """fc1 -> act -> fc2 -> drop (no dropout between, as in BERT; beit.py:40-58)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(fused.linear(self.fc2, self.act(fused.linear(self.fc1, x))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.,
                 window_size=None, attn_head_dim=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = attn_head_dim if attn_head_dim is not None else dim // num_heads
        all_head_dim = head_dim * num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            self.q_bias = self.v_bias = None
        if window_size:
            self.window_size = tuple(window_size)
            index, self.num_relative_distance = relative_position_index(self.window_size)
            self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
            self.register_buffer('relative_position_index', index)
        else:
            self.window_size = None
            self.relative_position_bias_table = None
            self.relative_position_index = None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(all_head_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def bias(self):
        """(heads, N, N) relative-position bias of this layer, or None."""
        if self.relative_position_bias_table is None:
            return None
        n = self.window_size[0] * self.window_size[1] + 1
        b = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        return b.permute(2, 0, 1)

    def forward(self, x, rel_pos_bias=None):
        B, N, C = x.shape
        qkv_bias = None
        if self.q_bias is not None:
            qkv_bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias, requires_grad=False), self.v_bias))
        packed = F.linear(x, self.qkv.weight, qkv_bias).reshape(B, N, 3, self.num_heads, -1)
        drop = self.attn_drop.p if self.training else 0.
        fast = x.is_cuda and packed.dtype == torch.bfloat16 and drop == 0.
        out = None
        if (fast and rel_pos_bias is None and self.relative_position_bias_table is not None
                and N == self.window_size[0] * self.window_size[1] + 1):
            # bf16 autocast: the MFMA kernels of csrc/attn_flash.hip with the relative position bias as an additive term
            # inside them, its two bf16 operands built straight from the table (no (heads, N, N) fp32 tensor)
            out = kernels.attention_relpos(packed, self.relative_position_bias_table, self.relative_position_index, self.scale)
        bias = None
        if out is None:
            bias = self.bias()
            if rel_pos_bias is not None:
                bias = rel_pos_bias if bias is None else bias + rel_pos_bias
            if fast:
                out = kernels.attention_bias(packed, bias, self.scale) if bias is not None else kernels.attention(packed, self.scale)
        if out is not None:
            return self.proj_drop(fused.linear(self.proj, out.reshape(B, N, -1)))
        qkv = packed.permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if x.is_cuda:
            mask = bias.unsqueeze(0).to(q.dtype) if bias is not None else None
            out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=self.scale,
                                                 dropout_p=self.attn_drop.p if self.training else 0.)
        else:                                   # the reference's op order (beit.py:120-144)
            attn = (q * self.scale) @ k.transpose(-2, -1)
            if bias is not None:
                attn = attn + bias.unsqueeze(0)
            out = self.attn_drop(attn.softmax(dim=-1)) @ v
        out = out.transpose(1, 2).reshape(B, N, -1)
        return self.proj_drop(fused.linear(self.proj, out))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 window_size=None, attn_head_dim=None, with_cp=False):
        super().__init__()
        self.with_cp = with_cp
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop, window_size=window_size,
                              attn_head_dim=attn_head_dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if init_values is not None:
            self.gamma_1 = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)
        else:
            self.gamma_1 = self.gamma_2 = None

    def _body(self, x, H, W, h=None, next_norm=None, rel_pos_bias=None):
        # x + drop_path(gamma_1 * attn(norm1(x))), x + drop_path(gamma_2 * mlp(norm2(x)))  (beit.py:176-184)
        if h is None:
            x, h = fused.layer_norm_keep(self.norm1, x)
        x, h = fused.residual_ln(x, self.attn(h, rel_pos_bias=rel_pos_bias), self.gamma_1, self.drop_path, self.norm2)
        f = self.mlp(h)
        if next_norm is not None:
            return fused.residual_ln(x, f, self.gamma_2, self.drop_path, next_norm)
        return fused.residual(x, f, self.gamma_2, self.drop_path)

    def forward(self, x, H, W, rel_pos_bias=None):
        if self.with_cp and x.requires_grad:
            return cp.checkpoint(lambda t: self._body(t, H, W, rel_pos_bias=rel_pos_bias), x, use_reentrant=False)
        return self._body(x, H, W, rel_pos_bias=rel_pos_bias)

    def forward_chain(self, x, H, W, h=None, next_norm=None):
        if self.with_cp and x.requires_grad:
            x = self.forward(x, H, W)
            return x if next_norm is None else fused.layer_norm_keep(next_norm, x)
        return self._body(x, H, W, h, next_norm)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.patch_shape = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.img_size, self.patch_size = img_size, patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x, **kwargs):
        out = fused.patch_embed(self.proj, x)              # bf16 autocast on the GPU: one GEMM on patch rows
        if out is not None:
            return out
        x = self.proj(x)
        Hp, Wp = x.shape[2], x.shape[3]
        return x.flatten(2).transpose(1, 2), Hp, Wp


class RelativePositionBias(nn.Module):
    """Bias table shared by all blocks (``use_shared_rel_pos_bias``; beit.py:259-292)."""

    def __init__(self, window_size, num_heads):
        super().__init__()
        self.window_size = tuple(window_size)
        index, self.num_relative_distance = relative_position_index(self.window_size)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        self.register_buffer('relative_position_index', index)

    def forward(self):
        n = self.window_size[0] * self.window_size[1] + 1
        b = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        return b.permute(2, 0, 1).contiguous()


class BEiT(nn.Module):
    def __init__(self, img_size=512, patch_size=16, in_chans=3, num_classes=80, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., hybrid_backbone=None, norm_layer=None, init_values=None,
                 use_checkpoint=False, use_abs_pos_emb=False, use_rel_pos_bias=True,
                 use_shared_rel_pos_bias=False, pretrained=None, with_cp=False):
        super().__init__()
        if hybrid_backbone is not None:
            raise NotImplementedError('hybrid CNN patch embedding is not part of the adapter path')
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.norm_layer = norm_layer
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.drop_path_rate = drop_path_rate
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim)) \
            if use_abs_pos_emb else None
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.rel_pos_bias = RelativePositionBias(self.patch_embed.patch_shape, num_heads) \
            if use_shared_rel_pos_bias else None
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.use_rel_pos_bias = use_rel_pos_bias
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  with_cp=with_cp, init_values=init_values,
                  window_size=self.patch_embed.patch_shape if use_rel_pos_bias else None)
            for i in range(depth)])
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.init_weights(pretrained)

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)
