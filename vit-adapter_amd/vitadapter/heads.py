"""Decode heads that consume the backbone's four maps: mmseg's ``UPerHead`` and ``FCNHead`` (SURVEY.md section 8 f-4,
BASELINE configs[3] "ViT-Adapter-L + UperNet head").

The reference only configures them (/root/reference/segmentation/configs/_base_/models/upernet_r50.py:17-41:
``UPerHead(in_channels, in_index=[0,1,2,3], pool_scales=(1,2,3,6), channels=512, dropout_ratio=0.1, norm SyncBN,
align_corners=False)`` + auxiliary ``FCNHead(in_index=2, channels=256, num_convs=1, concat_input=False)``; the L configs
set ``in_channels=[1024]*4``, ``num_classes=150``, upernet_augreg_adapter_large_512_160k_ade20k.py:30-31, and the BEiT one
``channels=1024``, upernet_beit_adapter_large_640_160k_ade20k_ss.py:34-41); the classes live in mmseg 0.20, which is not
part of the reference tree.  They are restated here from mmseg's published modules - pyramid pooling on the coarsest map,
top-down lateral sums, 3x3 FPN convolutions, all levels resized to the finest and fused by one 3x3 convolution, Dropout2d +
1x1 classifier; every ``ConvModule`` = bias-free convolution + (Sync)BatchNorm + ReLU.  PARITY UNPINNED against mmseg
(absent here); tests hold the modules to a functional re-evaluation of the same arithmetic from their state_dict.
Parameter names are mmseg's (``psp_modules.{i}.1.conv.weight``, ``bottleneck.bn.weight``, ``lateral_convs.{i}.conv.weight``,
``fpn_convs.{i}.*``, ``fpn_bottleneck.*``, ``conv_seg.*``; FCNHead: ``convs.0.conv.weight``, ``conv_seg.*``) so that a
reference checkpoint's ``decode_head.`` / ``auxiliary_head.`` entries load unchanged.
"""
import torch
import torch.nn.functional as F
from torch import nn


class ConvModule(nn.Module):
    """mmcv ConvModule(conv -> norm -> act): bias-free convolution when a norm follows."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, norm=nn.SyncBatchNorm, act=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, bias=norm is None)
        self.bn = norm(out_channels) if norm is not None else None
        self.activate = nn.ReLU(inplace=True) if act else None
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')     # mmcv ConvModule.init_weights
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.activate(x) if self.activate is not None else x


def resize(x, size, align_corners=False):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=align_corners)


class _Head(nn.Module):
    """What the two heads share of mmseg's BaseDecodeHead: input selection and the classifier."""

    def __init__(self, channels, num_classes, dropout_ratio, align_corners):
        super().__init__()
        self.channels, self.num_classes, self.align_corners = channels, num_classes, align_corners
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        nn.init.normal_(self.conv_seg.weight, mean=0, std=0.01)
        nn.init.zeros_(self.conv_seg.bias)

    def cls_seg(self, feat):
        if self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)


class UPerHead(_Head):
    def __init__(self, in_channels=(1024, 1024, 1024, 1024), in_index=(0, 1, 2, 3), pool_scales=(1, 2, 3, 6), channels=512,
                 dropout_ratio=0.1, num_classes=150, norm=nn.SyncBatchNorm, align_corners=False):
        super().__init__(channels, num_classes, dropout_ratio, align_corners)
        self.in_channels, self.in_index, self.pool_scales = list(in_channels), list(in_index), tuple(pool_scales)
        # PPM on the coarsest map
        self.psp_modules = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(s), ConvModule(self.in_channels[-1], channels, 1, norm=norm))
                                          for s in self.pool_scales])
        self.bottleneck = ConvModule(self.in_channels[-1] + len(self.pool_scales) * channels, channels, 3, padding=1, norm=norm)
        self.lateral_convs = nn.ModuleList([ConvModule(c, channels, 1, norm=norm) for c in self.in_channels[:-1]])
        self.fpn_convs = nn.ModuleList([ConvModule(channels, channels, 3, padding=1, norm=norm) for _ in self.in_channels[:-1]])
        self.fpn_bottleneck = ConvModule(len(self.in_channels) * channels, channels, 3, padding=1, norm=norm)

    def psp_forward(self, x):
        outs = [x] + [resize(m(x), x.shape[2:], self.align_corners) for m in self.psp_modules]
        return self.bottleneck(torch.cat(outs, dim=1))

    def forward(self, inputs):
        inputs = [inputs[i] for i in self.in_index]
        laterals = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)]
        laterals.append(self.psp_forward(inputs[-1]))
        n = len(laterals)
        for i in range(n - 1, 0, -1):                      # top-down path
            laterals[i - 1] = laterals[i - 1] + resize(laterals[i], laterals[i - 1].shape[2:], self.align_corners)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(n - 1)] + [laterals[-1]]
        for i in range(n - 1, 0, -1):
            outs[i] = resize(outs[i], outs[0].shape[2:], self.align_corners)
        return self.cls_seg(self.fpn_bottleneck(torch.cat(outs, dim=1)))


class FCNHead(_Head):
    """The auxiliary head of the UperNet configs: one 3x3 ConvModule on backbone map ``in_index``, then the classifier."""

    def __init__(self, in_channels=1024, in_index=2, channels=256, num_convs=1, concat_input=False, dropout_ratio=0.1,
                 num_classes=150, norm=nn.SyncBatchNorm, align_corners=False):
        super().__init__(channels, num_classes, dropout_ratio, align_corners)
        assert num_convs >= 1
        self.in_channels, self.in_index, self.concat_input = in_channels, in_index, concat_input
        self.convs = nn.Sequential(*[ConvModule(in_channels if i == 0 else channels, channels, 3, padding=1, norm=norm)
                                     for i in range(num_convs)])
        if concat_input:
            self.conv_cat = ConvModule(in_channels + channels, channels, 3, padding=1, norm=norm)

    def forward(self, inputs):
        x = inputs[self.in_index]
        out = self.convs(x)
        if self.concat_input:
            out = self.conv_cat(torch.cat([x, out], dim=1))
        return self.cls_seg(out)
