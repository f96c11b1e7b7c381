"""The deformable encoder of Mask2Former's pixel decoder on the gfx950 MSDA kernels (SURVEY.md section 8 f-3,
BASELINE configs[4]).

The reference builds it from mmcv / mmdet registry entries (``DetrTransformerEncoder`` of 6 ``BaseTransformerLayer``s,
``operation_order=('self_attn', 'norm', 'ffn', 'norm')``, attention ``MultiScaleDeformableAttention`` 256 ch / 8 heads /
3 levels / 4 points, ``FFN`` 256 -> 1024 -> 256 with ReLU, no dropout:
/root/reference/segmentation/configs/_base_/models/mask2former_beit.py:36-60) and drives it from
``MSDeformAttnPixelDecoder.forward`` (/root/reference/segmentation/mmseg_custom/models/plugins/
msdeformattn_pixel_decoder.py:160-242): the three coarsest backbone maps, projected to 256 channels and flattened from
low to high resolution, are the queries AND the values (Lq = S); ``query_pos`` = sine position + level embedding;
reference points = pixel centres of every level, repeated over the levels.

mmcv and mmdet are not part of the reference tree: the layer is restated from those call sites and mmcv 1.4's published
``BaseTransformerLayer`` / ``FFN`` (post-norm: ``x = norm(attn(x) + x)``, ``x = norm(x + ffn(x))``; the attention adds its
own identity, see mmcv_attention.py).  PARITY UNPINNED against mmcv; tests hold the stack to a plain PyTorch evaluation
of the same arithmetic on this repo's MSDA oracle.  Parameter names are mmcv's (``layers.{i}.attentions.0.*``,
``layers.{i}.ffns.0.layers.0.0.weight``, ``layers.{i}.ffns.0.layers.1.weight``, ``layers.{i}.norms.{0,1}.*``).
"""
import torch
from torch import nn

from . import fused
from .mmcv_attention import MultiScaleDeformableAttention


class FFN(nn.Module):
    """mmcv ``FFN(embed_dims, feedforward_channels, num_fcs=2, act ReLU, ffn_drop, add_identity=True)``."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, ffn_drop=0.0):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

    def forward(self, x, identity=None):
        h = fused.linear(self.layers[0][0], x)
        h = self.layers[0][2](torch.relu(h))
        h = self.layers[2](fused.linear(self.layers[1], h))
        return (x if identity is None else identity) + h


class DeformableEncoderLayer(nn.Module):
    """One ``BaseTransformerLayer`` with ``operation_order=('self_attn', 'norm', 'ffn', 'norm')``."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=3, num_points=4, feedforward_channels=1024, dropout=0.0):
        super().__init__()
        self.attentions = nn.ModuleList([MultiScaleDeformableAttention(
            embed_dims=embed_dims, num_heads=num_heads, num_levels=num_levels, num_points=num_points, dropout=dropout,
            batch_first=False)])
        self.ffns = nn.ModuleList([FFN(embed_dims, feedforward_channels, dropout)])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims), nn.LayerNorm(embed_dims)])

    def forward(self, query, query_pos=None, query_key_padding_mask=None, spatial_shapes=None, reference_points=None,
                level_start_index=None):
        query = self.attentions[0](query, query, query, identity=None, query_pos=query_pos,
                                   key_padding_mask=query_key_padding_mask, reference_points=reference_points,
                                   spatial_shapes=spatial_shapes, level_start_index=level_start_index)
        query = fused.layer_norm(self.norms[0], query)
        query = self.ffns[0](query)
        return fused.layer_norm(self.norms[1], query)


class MSDeformAttnEncoder(nn.Module):
    """``DetrTransformerEncoder(num_layers=6, transformerlayers=...)`` of the pixel decoder: (Lq, N, E) in and out."""

    def __init__(self, num_layers=6, embed_dims=256, num_heads=8, num_levels=3, num_points=4, feedforward_channels=1024,
                 dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([DeformableEncoderLayer(embed_dims, num_heads, num_levels, num_points,
                                                            feedforward_channels, dropout) for _ in range(num_layers)])
        self.embed_dims = embed_dims
        self.init_weights()

    def init_weights(self):
        """msdeformattn_pixel_decoder.py:143-158: Xavier on every matrix, then the attention's own init."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        for layer in self.layers:
            layer.attentions[0].init_weights()

    def forward(self, query, query_pos=None, query_key_padding_mask=None, spatial_shapes=None, reference_points=None,
                level_start_index=None, **kwargs):
        with fused.forward_epoch(self):
            for layer in self.layers:
                query = layer(query, query_pos, query_key_padding_mask, spatial_shapes, reference_points, level_start_index)
        return query


def encoder_inputs(level_shapes, batch, embed_dims, device, seed=0, dtype=torch.float32):
    """Synthetic inputs of the encoder as MSDeformAttnPixelDecoder.forward builds them (:176-228): levels from low to high
    resolution, queries (Lq, N, E), query_pos (Lq, N, E), pixel-centre reference points (N, Lq, L, 2), int64 geometry."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    L = len(level_shapes)
    Lq = sum(h * w for h, w in level_shapes)
    query = torch.randn(Lq, batch, embed_dims, generator=g).to(device=device, dtype=dtype)
    pos = torch.randn(Lq, batch, embed_dims, generator=g).to(device=device, dtype=dtype)
    pts = []
    for h, w in level_shapes:
        ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
        xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
        gy, gx = torch.meshgrid(ys, xs, indexing='ij')
        pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    ref = torch.cat(pts, 0)[None, :, None].repeat(batch, 1, L, 1).to(device)
    shapes = torch.as_tensor(level_shapes, dtype=torch.long, device=device)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return query, pos, ref, shapes, lsi
