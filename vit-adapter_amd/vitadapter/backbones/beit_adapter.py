"""BEiTAdapter: the adapter around a BEiT trunk.

Interface mirror of /root/reference/segmentation/mmseg_custom/models/backbones/beit_adapter.py:20-141
(constructor arguments, parameter names, four-scale output); the class token travels through the
ViT blocks and is split off again around every injector / extractor
(``InteractionBlockWithCls``, adapter_modules.py:194-232).  BASELINE config 4's published model
(seg/configs/ade20k/upernet_beit_adapter_large_640_160k_ade20k_ss.py:13-33).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from ops.modules import MSDeformAttn

from .. import fused, spm_nhwc
from .adapter_modules import InteractionBlockWithCls, SpatialPriorModule, deform_inputs
from .beit import BEiT
from .vit_adapter import ViTAdapter


class BEiTAdapter(BEiT):
    def __init__(self, pretrain_size=224, conv_inplane=64, n_points=4, deform_num_heads=6, init_values=0.,
                 cffn_ratio=0.25, deform_ratio=1.0, with_cffn=True, interaction_indexes=None,
                 add_vit_feature=True, with_cp=False, *args, **kwargs):
        super().__init__(init_values=init_values, with_cp=with_cp, *args, **kwargs)
        self.num_block = len(self.blocks)
        self.pretrain_size = (pretrain_size, pretrain_size)
        self.flags = [i for i in range(-1, self.num_block, self.num_block // 4)][1:]
        self.interaction_indexes = interaction_indexes
        self.add_vit_feature = add_vit_feature
        self._pos_resize_cache = {}
        embed_dim = self.embed_dim

        self.level_embed = nn.Parameter(torch.zeros(3, embed_dim))
        self.spm = SpatialPriorModule(inplanes=conv_inplane, embed_dim=embed_dim, with_cp=False)
        last = len(interaction_indexes) - 1
        self.interactions = nn.Sequential(*[
            InteractionBlockWithCls(dim=embed_dim, num_heads=deform_num_heads, n_points=n_points,
                                    init_values=init_values, drop_path=self.drop_path_rate,
                                    norm_layer=self.norm_layer, with_cffn=with_cffn, cffn_ratio=cffn_ratio,
                                    deform_ratio=deform_ratio, extra_extractor=(i == last), with_cp=with_cp)
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

    # initialisation rules and pos-embed resize are the ViT adapter's (beit_adapter.py:61-90 repeats them)
    _init_weights = ViTAdapter._init_weights
    _bicubic_matrix = ViTAdapter._bicubic_matrix
    _get_pos_embed = ViTAdapter._get_pos_embed
    _add_level_embed = ViTAdapter._add_level_embed

    def _init_deform_weights(self, m):
        if isinstance(m, MSDeformAttn):
            m._reset_parameters()

    def forward(self, x):
        with fused.forward_epoch(self):        # bf16 copies of the Linear weights for THIS forward: one launch
            return self._forward(x)

    def _forward(self, x):
        deform_inputs1, deform_inputs2 = deform_inputs(x)

        # fused tail: the biases of spm.fc1 and self.up reach norm1 as a per-channel shift
        fold = self.add_vit_feature and fused.tail_takes_conv_bias(self.norm1, x)
        if fold and spm_nhwc.usable(self.spm, x) and not (self.spm.with_cp and x.requires_grad):
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
        cls = self.cls_token.expand(bs, -1, -1)
        if self.pos_embed is not None:
            x = x + self._get_pos_embed(self.pos_embed[:, 1:] if self.pos_embed.shape[1] == n + 1
                                        else self.pos_embed, H, W)
        x = self.pos_drop(x)

        outs = []
        for i, layer in enumerate(self.interactions):
            lo, hi = self.interaction_indexes[i][0], self.interaction_indexes[i][-1]
            x, c, cls = layer(x, c, cls, self.blocks[lo:hi + 1], deform_inputs1, deform_inputs2, H, W)
            outs.append(fused.tokens_to_maps(x, [(H, W)])[0])

        c2, c3, c4 = fused.tokens_to_maps(c, [(H * 2, W * 2), (H, W), (H // 2, W // 2)])
        if self.add_vit_feature:
            x1, x2, x3, x4 = outs
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


def register_beit_adapter(registry=None, name='BEiTAdapter', force=True, flavour='seg'):
    """Register into mmseg's (flavour 'seg') or mmdet's ('det': the windowed trunk without class token,
    backbones/beit_det.py) BACKBONES under the reference's name."""
    if flavour == 'det':
        from .beit_det import BEiTAdapter as cls
    else:
        cls = BEiTAdapter
    if registry is None:
        if flavour == 'det':
            from mmdet.models.builder import BACKBONES as registry
        else:
            from mmseg.models.builder import BACKBONES as registry
    registry.register_module(name=name, force=force, module=cls)
    return cls
