"""BEiTAdapter, detection flavour ("det windowed variant").

Interface mirror of /root/reference/detection/mmdet_custom/models/backbones/base/beit.py:94-440 (trunk) and
/root/reference/detection/mmdet_custom/models/backbones/beit_adapter.py:20-145 (adapter): a BEiT trunk WITHOUT class
token whose blocks attend either inside ``window_size`` x ``window_size`` windows (the token grid zero-padded BEFORE the
qkv projection, base/beit.py:175-195 - unlike the ViT flavour, which pads q, k, v after it) or over the whole
``window_size``^2 grid, every block with its own relative position bias table of (2 w - 1)^2 rows; parameter names and
shapes equal the reference's.  Used by the htc++ configs (window_attn / window_size lists, 14 and 56).

Under bf16 autocast on the GPU the attention runs on the MFMA kernels of csrc/attn_flash.hip with the bias as an
additive term (kernels.attention_relpos): the windows of a block are one batch of sequences for them, because here the
projection output already is per-window contiguous.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp
from ops.modules import MSDeformAttn

from .. import fused, kernels, spm_nhwc
from .adapter_modules import InteractionBlock, SpatialPriorModule, deform_inputs
from .beit import Mlp, PatchEmbed
from .vit import DropPath
from .vit_adapter import ViTAdapter


def window_relative_position_index(window_size):
    """(w*w, w*w) index into the (2w-1)^2-row bias table (base/beit.py:122-134)."""
    ys, xs = torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing='ij')
    coords = torch.stack([ys.flatten(), xs.flatten()])
    rel = coords[:, :, None] - coords[:, None, :]
    return (rel[0] + window_size - 1) * (2 * window_size - 1) + (rel[1] + window_size - 1)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., window_size=None,
                 attn_head_dim=None, windowed=False):
        super().__init__()
        self.num_heads = num_heads
        self.windowed = windowed
        head_dim = attn_head_dim if attn_head_dim is not None else dim // num_heads
        all_head_dim = head_dim * num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            self.q_bias = self.v_bias = None
        self.window_size = window_size
        self.num_relative_distance = (2 * window_size - 1) * (2 * window_size - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        self.register_buffer('relative_position_index', window_relative_position_index(window_size))
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(all_head_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def _attend(self, x, rel_pos_bias):
        """(B', N, C) with N == window_size^2 -> (B', N, C)   (base/beit.py:141-173)."""
        B, N, C = x.shape
        qkv_bias = None
        if self.q_bias is not None:
            qkv_bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias, requires_grad=False), self.v_bias))
        packed = F.linear(x, self.qkv.weight, qkv_bias).reshape(B, N, 3, self.num_heads, -1)
        drop = self.attn_drop.p if self.training else 0.
        out = None
        if x.is_cuda and packed.dtype == torch.bfloat16 and drop == 0. and rel_pos_bias is None:
            out = kernels.attention_relpos(packed, self.relative_position_bias_table, self.relative_position_index, self.scale)
        if out is None:                      # the reference's op order
            q, k, v = packed.permute(2, 0, 3, 1, 4).unbind(0)
            attn = (q * self.scale) @ k.transpose(-2, -1)
            n = self.window_size * self.window_size
            bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
            attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
            if rel_pos_bias is not None:
                attn = attn + rel_pos_bias
            attn = self.attn_drop(attn.softmax(dim=-1))
            out = (attn @ v).transpose(1, 2)
        return self.proj_drop(fused.linear(self.proj, out.reshape(B, N, -1)))

    def forward(self, x, H, W, rel_pos_bias=None):
        if not self.windowed:
            return self._attend(x, rel_pos_bias)
        B, L, C = x.shape
        w = self.window_size
        Hp, Wp = math.ceil(H / w) * w, math.ceil(W / w) * w
        x = F.pad(x.view(B, H, W, C), [0, 0, 0, Wp - W, 0, Hp - H])
        x = x.view(B, Hp // w, w, Wp // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, w * w, C)       # window_partition
        x = self._attend(x, rel_pos_bias)
        x = x.view(B, Hp // w, Wp // w, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)       # window_reverse
        return x[:, :H, :W, :].reshape(B, H * W, C)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm, window_size=None, windowed=False,
                 attn_head_dim=None, with_cp=False):
        super().__init__()
        self.with_cp = with_cp
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, window_size=window_size, attn_head_dim=attn_head_dim, windowed=windowed)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if init_values is not None:
            self.gamma_1 = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)
        else:
            self.gamma_1 = self.gamma_2 = None

    def _body(self, x, H, W, h=None, next_norm=None, rel_pos_bias=None):
        # x + drop_path(gamma_1 * attn(norm1(x))), x + drop_path(gamma_2 * mlp(norm2(x)))  (base/beit.py:224-234)
        if h is None:
            x, h = fused.layer_norm_keep(self.norm1, x)
        x, h = fused.residual_ln(x, self.attn(h, H, W, rel_pos_bias=rel_pos_bias), self.gamma_1, self.drop_path, self.norm2)
        f = self.mlp(h)
        if next_norm is not None:
            return fused.residual_ln(x, f, self.gamma_2, self.drop_path, next_norm)
        return fused.residual(x, f, self.gamma_2, self.drop_path)

    def forward(self, x, H, W, rel_pos_bias=None):
        if self.with_cp and x.requires_grad:
            return cp.checkpoint(lambda t: self._body(t, H, W, rel_pos_bias=rel_pos_bias), x, use_reentrant=False)
        return self._body(x, H, W, rel_pos_bias=rel_pos_bias)

    def forward_chain(self, x, H, W, h=None, next_norm=None):
        """The block with its last residual update fused to the next block's first LayerNorm (vit.run_blocks)."""
        if self.with_cp and x.requires_grad:
            x = self.forward(x, H, W)
            return x if next_norm is None else fused.layer_norm_keep(next_norm, x)
        return self._body(x, H, W, h, next_norm)


class BEiT(nn.Module):
    def __init__(self, img_size=512, patch_size=16, in_chans=3, num_classes=80, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 hybrid_backbone=None, norm_layer=None, init_values=None, use_checkpoint=False, use_abs_pos_emb=False,
                 use_rel_pos_bias=True, use_shared_rel_pos_bias=False, pretrained=None, with_cp=False, window_attn=False,
                 window_size=14):
        super().__init__()
        if hybrid_backbone is not None:
            raise NotImplementedError('hybrid CNN patch embedding is not part of the adapter path')
        if use_shared_rel_pos_bias:
            raise NotImplementedError('the det flavour is configured with per-block tables (use_shared_rel_pos_bias=False)')
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.norm_layer = norm_layer
        self.num_features = self.embed_dim = embed_dim
        self.drop_path_rate = drop_path_rate
        window_attn = [window_attn] * depth if not isinstance(window_attn, list) else window_attn
        window_size = [window_size] * depth if not isinstance(window_size, list) else window_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim)) if use_abs_pos_emb else None
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.rel_pos_bias = None
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.use_rel_pos_bias = use_rel_pos_bias
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, with_cp=with_cp,
                  init_values=init_values, windowed=window_attn[i], window_size=window_size[i])
            for i in range(depth)])
        if self.pos_embed is not None:
            nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.apply(self._init_weights)
        self.init_weights(pretrained)

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False, flavour='det')

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


class BEiTAdapter(BEiT):
    def __init__(self, pretrain_size=224, conv_inplane=64, n_points=4, deform_num_heads=6, init_values=0., cffn_ratio=0.25,
                 deform_ratio=1.0, with_cffn=True, interaction_indexes=None, add_vit_feature=True, version='new',
                 with_cp=False, *args, **kwargs):
        super().__init__(init_values=init_values, with_cp=with_cp, *args, **kwargs)
        self.version = version
        self.num_block = len(self.blocks)
        self.pretrain_size = (pretrain_size, pretrain_size)
        self.interaction_indexes = interaction_indexes
        self.add_vit_feature = add_vit_feature
        embed_dim = self.embed_dim
        self.level_embed = nn.Parameter(torch.zeros(3, embed_dim))
        self.spm = SpatialPriorModule(inplanes=conv_inplane, embed_dim=embed_dim)
        self.interactions = nn.Sequential(*[
            InteractionBlock(dim=embed_dim, num_heads=deform_num_heads, n_points=n_points, init_values=init_values,
                             drop_path=self.drop_path_rate, norm_layer=self.norm_layer, with_cffn=with_cffn,
                             cffn_ratio=cffn_ratio, deform_ratio=deform_ratio,
                             extra_extractor=(i == len(interaction_indexes) - 1), with_cp=with_cp)
            for i in range(len(interaction_indexes))])
        self.up = nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2)
        self.norm1 = nn.SyncBatchNorm(embed_dim)
        self.norm2 = nn.SyncBatchNorm(embed_dim)
        self.norm3 = nn.SyncBatchNorm(embed_dim)
        self.norm4 = nn.SyncBatchNorm(embed_dim)
        self.up.apply(self._init_weights)
        self.spm.apply(self._init_weights)
        self.interactions.apply(self._init_weights)
        self.apply(self._init_deform_weights)
        nn.init.normal_(self.level_embed)

    _init_weights = ViTAdapter._init_weights

    def _init_deform_weights(self, m):
        if isinstance(m, MSDeformAttn):
            m._reset_parameters()

    def _get_pos_embed(self, pos_embed, H, W):
        pos_embed = pos_embed.reshape(1, self.pretrain_size[0] // 16, self.pretrain_size[1] // 16, -1).permute(0, 3, 1, 2)
        return F.interpolate(pos_embed, size=(H, W), mode='bicubic', align_corners=False).reshape(1, -1, H * W).permute(0, 2, 1)

    def forward(self, x):
        with fused.forward_epoch(self):
            return self._forward(x)

    def _forward(self, x):
        deform_inputs1, deform_inputs2 = deform_inputs(x)
        fold = self.add_vit_feature and fused.tail_takes_conv_bias(self.norm1, x)
        if fold and spm_nhwc.usable(self.spm, x) and not (self.spm.with_cp and x.requires_grad):
            c1, c = spm_nhwc.forward(self.spm, x, self.level_embed)
        else:
            c1, c2, c3, c4 = self.spm(x, bias_free_c1=fold)
            c = torch.cat([c2 + self.level_embed[0], c3 + self.level_embed[1], c4 + self.level_embed[2]], dim=1)
        x, H, W = self.patch_embed(x)
        bs, n, dim = x.shape
        if self.pos_embed is not None:
            x = x + self._get_pos_embed(self.pos_embed, H, W)
        x = self.pos_drop(x)
        outs = []
        for i, layer in enumerate(self.interactions):
            lo, hi = self.interaction_indexes[i][0], self.interaction_indexes[i][-1]
            x, c = layer(x, c, self.blocks[lo:hi + 1], deform_inputs1, deform_inputs2, H, W)
            if self.version == 'old':
                outs.append(fused.tokens_to_maps(x, [(H, W)])[0])
        c2, c3, c4 = fused.tokens_to_maps(c, [(H * 2, W * 2), (H, W), (H // 2, W // 2)])
        if self.add_vit_feature:
            if self.version == 'old':
                x1, x2, x3, x4 = outs
            else:
                x1 = x2 = x3 = x4 = fused.tokens_to_maps(x, [(H, W)])[0]
            c4 = c4 + fused.halve(x4)
            up = fused.up_from_tokens(self.up, c[:, :4 * H * W], 2 * H, 2 * W, c1 if c1.dtype == torch.bfloat16 else None) if fold else None
            if up is not None and c1.dtype == torch.bfloat16:
                c1 = None
            if up is None:
                up = F.conv_transpose2d(c2, self.up.weight, None, stride=2) if fold else self.up(c2)
            shift = self.spm.fc1.bias + self.up.bias if fold else None
            return [fused.bn_tail(self.norm1, up, c1, x1, 4, shift), fused.bn_tail(self.norm2, c2, None, x2, 2),
                    fused.bn_tail(self.norm3, c3, None, x3, 1), self.norm4(c4)]
        c1 = self.up(c2) + c1
        return [self.norm1(c1), self.norm2(c2), self.norm3(c3), self.norm4(c4)]
