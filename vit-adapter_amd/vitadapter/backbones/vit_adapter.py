"""ViTAdapter backbone: ViT trunk + spatial prior module + injector/extractor interactions,
returning the 4-scale feature pyramid [1/4, 1/8, 1/16, 1/32], each with embed_dim channels.

Mirror of the reference's two classes of the same name:
  seg  /root/reference/segmentation/mmseg_custom/models/backbones/vit_adapter.py:19-137
       (ctor has ``pretrained`` / ``with_cp``; the pyramid adds the per-stage ViT outputs)
  det  /root/reference/detection/mmdet_custom/models/backbones/vit_adapter.py:19-132
       (no ``with_cp`` of its own; all four pyramid levels add resized copies of the FINAL x)
Constructor keywords, sub-module names (=> state_dict keys), init rules and forward arithmetic
are the reference's; ``flavour`` selects which of the two forwards is computed.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from ops.modules import MSDeformAttn

from .. import fused, spm_nhwc
from .adapter_modules import InteractionBlock, SpatialPriorModule, deform_inputs
from .vit import TIMMVisionTransformer


class ViTAdapter(TIMMVisionTransformer):
    flavour = 'seg'

    def __init__(self, pretrain_size=224, num_heads=12, conv_inplane=64, n_points=4,
                 deform_num_heads=6, init_values=0., interaction_indexes=None, with_cffn=True,
                 cffn_ratio=0.25, deform_ratio=1.0, add_vit_feature=True, pretrained=None,
                 use_extra_extractor=True, with_cp=False, flavour=None, *args, **kwargs):
        super().__init__(num_heads=num_heads, pretrained=pretrained, with_cp=with_cp,
                         *args, **kwargs)
        if flavour is not None:
            if flavour not in ('seg', 'det'):
                raise ValueError("flavour must be 'seg' or 'det'")
            self.flavour = flavour
        self.cls_token = None
        self._pos_resize_cache = {}
        self.num_block = len(self.blocks)
        self.pretrain_size = (pretrain_size, pretrain_size)
        self.interaction_indexes = interaction_indexes
        self.add_vit_feature = add_vit_feature
        embed_dim = self.embed_dim

        self.level_embed = nn.Parameter(torch.zeros(3, embed_dim))
        self.spm = SpatialPriorModule(inplanes=conv_inplane, embed_dim=embed_dim, with_cp=False)
        last = len(interaction_indexes) - 1
        self.interactions = nn.Sequential(*[
            InteractionBlock(dim=embed_dim, num_heads=deform_num_heads, n_points=n_points,
                             init_values=init_values, drop_path=self.drop_path_rate,
                             norm_layer=self.norm_layer, with_cffn=with_cffn,
                             cffn_ratio=cffn_ratio, deform_ratio=deform_ratio,
                             extra_extractor=(i == last and use_extra_extractor),
                             with_cp=with_cp)
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

    # --- initialisation rules of the reference (vit_adapter.py:61-74, 83-85) ------------------
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def _init_deform_weights(self, m):
        if isinstance(m, MSDeformAttn):
            m._reset_parameters()

    def _bicubic_matrix(self, H, W, device):
        """(H*W, ph*pw) matrix A with  A @ pos == bicubic_resize(pos)  (align_corners=False).

        Bicubic resampling is a fixed linear map of the ph*pw source positions, so it is built
        once per output size by resizing the identity basis with the very F.interpolate call the
        reference uses (vit_adapter.py:76-81) and then applied as one small GEMM.  Same values as
        the reference; the backward becomes A^T @ grad instead of PyTorch's atomic scatter kernel
        (measured 55.7 ms of a 163 ms step for ViT-B at 1024x1024, profiles/r01_*)."""
        key = (H, W, str(device))
        mat = self._pos_resize_cache.get(key)
        if mat is None:
            ph, pw = self.pretrain_size[0] // 16, self.pretrain_size[1] // 16
            eye = torch.eye(ph * pw, dtype=torch.float32, device=device).view(ph * pw, 1, ph, pw)
            mat = F.interpolate(eye, size=(H, W), mode='bicubic', align_corners=False)
            mat = mat.view(ph * pw, H * W).t().contiguous()
            if len(self._pos_resize_cache) > 16:
                self._pos_resize_cache.clear()
            self._pos_resize_cache[key] = mat
        return mat

    def _get_pos_embed(self, pos_embed, H, W):
        """pos_embed (1, ph*pw, E) -> (1, H*W, E), bicubic (vit_adapter.py:76-81)."""
        ph, pw = self.pretrain_size[0] // 16, self.pretrain_size[1] // 16
        if (H, W) == (ph, pw):
            return pos_embed
        with torch.autocast('cuda', enabled=False):
            A = self._bicubic_matrix(H, W, pos_embed.device)
            return (A @ pos_embed[0].float()).unsqueeze(0)

    def _add_level_embed(self, c2, c3, c4):
        return c2 + self.level_embed[0], c3 + self.level_embed[1], c4 + self.level_embed[2]

    def forward(self, x):
        with fused.forward_epoch(self):        # bf16 copies of the Linear weights for THIS forward: one launch
            return self._forward(x)

    def _forward(self, x):
        deform_inputs1, deform_inputs2 = deform_inputs(x)

        # fused tail: the biases of spm.fc1 and self.up reach norm1 as a per-channel shift
        fold = self.add_vit_feature and fused.tail_takes_conv_bias(self.norm1, x)
        if fold and spm_nhwc.usable(self.spm, x) and not (self.spm.with_cp and x.requires_grad):
            # the whole module on NHWC bf16 with its own convolution kernels; its stride-8/16/32 maps are the token rows
            c1, c = spm_nhwc.forward(self.spm, x, self.level_embed)
        elif fused.ENABLED['maps'] and fused.ENABLED['maps_in'] and x.is_cuda:
            # c2..c4 leave the SPM as bias-free maps; bias + level embedding are added while the token
            # sequence is laid out (one pass per map instead of bias add, level add and cat)
            c1, m2, m3, m4 = self.spm(x, bias_free_c1=fold, raw_maps=True)
            c = fused.maps_to_tokens([m2, m3, m4], [f.bias + self.level_embed[i] for i, f in
                                                   enumerate((self.spm.fc2, self.spm.fc3, self.spm.fc4))])
        else:
            c1, c2, c3, c4 = self.spm(x, bias_free_c1=fold)
            c2, c3, c4 = self._add_level_embed(c2, c3, c4)
            c = torch.cat([c2, c3, c4], dim=1)

        x, H, W = self.patch_embed(x)
        bs, n, dim = x.shape
        x = self.pos_drop(x + self._get_pos_embed(self.pos_embed[:, 1:], H, W))

        stage_maps = []
        for i, layer in enumerate(self.interactions):
            lo, hi = self.interaction_indexes[i][0], self.interaction_indexes[i][-1]
            x, c = layer(x, c, self.blocks[lo:hi + 1], deform_inputs1, deform_inputs2, H, W)
            if self.flavour == 'seg':
                stage_maps.append(fused.tokens_to_maps(x, [(H, W)])[0])

        c2, c3, c4 = fused.tokens_to_maps(c, [(H * 2, W * 2), (H, W), (H // 2, W // 2)])
        if self.add_vit_feature:
            if self.flavour == 'seg':
                x1, x2, x3, x4 = stage_maps
            else:
                x1 = x2 = x3 = x4 = fused.tokens_to_maps(x, [(H, W)])[0]
            # f1 = norm1(up(c2) + c1 + interp(x1, 4)), f2 = norm2(c2 + interp(x2, 2)), f3 = norm3(c3 + x3):
            # sum, upsampling and batch norm in one pair of passes (csrc/tail_ops.hip); off the bf16
            # GPU path fused.bn_tail evaluates exactly the reference expression
            c4 = c4 + fused.halve(x4)
            # GEMM form on the token rows, c1 summed in by its interleave pass (one bf16 operand for the tail instead of two)
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


class ViTAdapterSeg(ViTAdapter):
    """mmseg_custom flavour."""
    flavour = 'seg'


class ViTAdapterDet(ViTAdapter):
    """mmdet_custom flavour: its ctor takes no ``with_cp`` / ``pretrained`` of its own
    (they travel in **kwargs to the trunk, det vit_adapter.py:21-25)."""
    flavour = 'det'


# BASELINE configs as constructor keyword presets.
PRESETS = {
    # seg/configs/ade20k/upernet_deit_adapter_tiny_512_160k_ade20k.py:10-26
    'tiny_seg': dict(flavour='seg', patch_size=16, embed_dim=192, depth=12, num_heads=3,
                     mlp_ratio=4, drop_path_rate=0.1, conv_inplane=64, n_points=4,
                     deform_num_heads=6, cffn_ratio=0.25, deform_ratio=1.0,
                     interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]],
                     window_attn=[False] * 12, window_size=[None] * 12),
    # det/configs/mask_rcnn/mask_rcnn_deit_adapter_base_fpn_3x_coco.py:10-29
    'base_det': dict(flavour='det', patch_size=16, embed_dim=768, depth=12, num_heads=12,
                     mlp_ratio=4, drop_path_rate=0.3, conv_inplane=64, n_points=4,
                     deform_num_heads=12, cffn_ratio=0.25, deform_ratio=0.5,
                     interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]],
                     window_attn=[True, True, False] * 4, window_size=[14, 14, None] * 4),
    # seg/configs/ade20k/upernet_augreg_adapter_large_512_160k_ade20k.py (ViT-L AugReg shape)
    'large_seg': dict(flavour='seg', patch_size=16, embed_dim=1024, depth=24, num_heads=16,
                      mlp_ratio=4, drop_path_rate=0.4, conv_inplane=64, n_points=4,
                      deform_num_heads=16, cffn_ratio=0.25, deform_ratio=0.5, with_cp=True,
                      interaction_indexes=[[0, 5], [6, 11], [12, 17], [18, 23]],
                      window_attn=[False] * 24, window_size=[None] * 24),
    # seg/configs/ade20k/upernet_beit_adapter_large_640_160k_ade20k_ss.py:13-33 (BASELINE configs[3] as published):
    # built by BEiTAdapter (class token, relative position bias, layer scale), see build_preset
    'beit_large_seg': dict(flavour='seg', beit=True, img_size=640, patch_size=16, embed_dim=1024, depth=24, num_heads=16,
                           mlp_ratio=4, qkv_bias=True, use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=1e-6,
                           drop_path_rate=0.3, conv_inplane=64, n_points=4, deform_num_heads=16, cffn_ratio=0.25,
                           deform_ratio=0.5, with_cp=True, interaction_indexes=[[0, 5], [6, 11], [12, 17], [18, 23]]),
}


def build_preset(name, **overrides):
    kw = dict(PRESETS[name])
    kw.update(overrides)
    if kw.pop('beit', False):
        from .beit_adapter import BEiTAdapter
        kw.pop('flavour', None)
        return BEiTAdapter(**kw)
    return ViTAdapter(**kw)


def register_backbones(registry=None, flavour='seg', name='ViTAdapter', force=True):
    """Register this backbone into an OpenMMLab ``BACKBONES`` registry under the reference's
    name so that reference configs (``backbone=dict(type='ViTAdapter', ...)``) build it
    unchanged.  ``registry`` defaults to mmseg's / mmdet's (imported lazily)."""
    if registry is None:
        if flavour == 'seg':
            from mmseg.models.builder import BACKBONES as registry
        else:
            from mmdet.models.builder import BACKBONES as registry
    cls = ViTAdapterSeg if flavour == 'seg' else ViTAdapterDet
    registry.register_module(name=name, force=force, module=cls)
    return cls
