"""Adapter side of ViT-Adapter: spatial prior module, injector / extractor cross attention on
multi-scale deformable attention, convolutional FFN, interaction blocks.

Behavioural mirror (same sub-module names => same state_dict keys, same arithmetic) of
/root/reference/segmentation/mmseg_custom/models/backbones/adapter_modules.py:13-296 (the
detection copy, /root/reference/detection/mmdet_custom/models/backbones/adapter_modules.py, is
the same code minus InteractionBlockWithCls).

Host-side differences that do not change results: the sampling geometry (reference points,
spatial shapes, level start indices) is a pure function of the input size and is cached per
(H, W, device) instead of being rebuilt with ~20 tiny kernels and two H2D copies every forward
(adapter_modules.py:28-47).
"""
from functools import partial

import torch
import torch.nn as nn
import torch.utils.checkpoint as cp
from ops.modules import MSDeformAttn

from .. import fused
from .vit import DropPath, run_blocks


def get_reference_points(spatial_shapes, device):
    """Pixel-centre grid of each (H, W) level, normalised to (0, 1): (1, sum HW, 1, 2) as (x, y)."""
    per_level = []
    for H_, W_ in spatial_shapes:
        ys = torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device) / H_
        xs = torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device) / W_
        gy, gx = torch.meshgrid(ys, xs, indexing='ij')
        per_level.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1)[None])
    return torch.cat(per_level, 1)[:, :, None]


def _level_start_index(spatial_shapes):
    return torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))


_GEOMETRY_CACHE = {}


def deform_inputs(x):
    """-> (inputs for the injector, inputs for the extractor), each
    [reference_points, spatial_shapes (L,2) int64, level_start_index (L,) int64].

    Injector: ViT tokens (stride 16) query the three SPM maps (strides 8/16/32).
    Extractor: the SPM tokens (three levels) query the single stride-16 ViT map."""
    _, _, h, w = x.shape
    key = (h, w, str(x.device))
    hit = _GEOMETRY_CACHE.get(key)
    if hit is not None:
        return hit
    pyramid = [(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)]
    vit = [(h // 16, w // 16)]
    out = []
    for value_shapes, query_shapes in ((pyramid, vit), (vit, pyramid)):
        shapes = torch.as_tensor(value_shapes, dtype=torch.long, device=x.device)
        out.append([get_reference_points(query_shapes, x.device), shapes,
                    _level_start_index(shapes)])
    if len(_GEOMETRY_CACHE) > 64:
        _GEOMETRY_CACHE.clear()
    _GEOMETRY_CACHE[key] = tuple(out)
    return _GEOMETRY_CACHE[key]


class DWConv(nn.Module):
    """Depthwise 3x3 over the three token maps (strides 8 / 16 / 32) that are concatenated along
    the token axis as 16n + 4n + n tokens; the same filter is shared by the three maps."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W):
        y = fused.dwconv_tokens(self.dwconv, x, H, W)      # one HIP kernel on the token layout
        if y is not None:
            return y
        B, N, C = x.shape
        n = N // 21
        outs = []
        for lo, hi, (h, w) in ((0, 16 * n, (H * 2, W * 2)), (16 * n, 20 * n, (H, W)),
                               (20 * n, N, (H // 2, W // 2))):
            m = x[:, lo:hi, :].transpose(1, 2).reshape(B, C, h, w).contiguous()
            outs.append(self.dwconv(m).flatten(2).transpose(1, 2))
        return torch.cat(outs, dim=1)


class ConvFFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, H, W):
        x = self.drop(self.act(self.dwconv(fused.linear(self.fc1, x), H, W)))
        return self.drop(fused.linear(self.fc2, x))


class Extractor(nn.Module):
    """c <- c + MSDA(LN c, LN x);  c <- c + DropPath(ConvFFN(LN c))."""

    def __init__(self, dim, num_heads=6, n_points=4, n_levels=1, deform_ratio=1.0,
                 with_cffn=True, cffn_ratio=0.25, drop=0., drop_path=0.,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), with_cp=False):
        super().__init__()
        self.query_norm = norm_layer(dim)
        self.feat_norm = norm_layer(dim)
        self.attn = MSDeformAttn(d_model=dim, n_levels=n_levels, n_heads=num_heads,
                                 n_points=n_points, ratio=deform_ratio)
        self.with_cffn = with_cffn
        self.with_cp = with_cp
        if with_cffn:
            self.ffn = ConvFFN(in_features=dim, hidden_features=int(dim * cffn_ratio), drop=drop)
            self.ffn_norm = norm_layer(dim)
            self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index, H, W, keep_feat=False,
                query_normed=None):
        """keep_feat: also return ``feat`` as it leaves the feat_norm node - a caller that goes on using
        THAT tensor (x feeds further extractors and the next interaction) gets the sum of its gradients
        formed inside the LayerNorm backward kernel instead of by a separate add per consumer.
        query_normed: ``self.query_norm(query)`` when the caller already has it (InteractionBlock shares
        the statistics of c between the injector's feat_norm and this query_norm)."""
        def body(query, feat, qn=None):
            if qn is None:
                query, qn = fused.layer_norm_keep(self.query_norm, query)
            feat, fn = fused.layer_norm_keep(self.feat_norm, feat, fan_out=True)
            attn = self.attn(qn, reference_points, fn, spatial_shapes, level_start_index, None)
            if self.with_cffn:
                query, qn = fused.residual_ln(query, attn, None, None, self.ffn_norm)
                out = fused.residual(query, self.ffn(qn, H, W), None, self.drop_path)
            else:
                out = fused.residual(query, attn)
            return out, feat

        if self.with_cp and query.requires_grad:
            out, feat = cp.checkpoint(body, query, feat, query_normed, use_reentrant=False)
        else:
            out, feat = body(query, feat, query_normed)
        return (out, feat) if keep_feat else out


class Injector(nn.Module):
    """x <- x + gamma * MSDA(LN x, LN c); gamma starts at ``init_values`` (0: identity)."""

    def __init__(self, dim, num_heads=6, n_points=4, n_levels=1, deform_ratio=1.0,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=0., with_cp=False):
        super().__init__()
        self.with_cp = with_cp
        self.query_norm = norm_layer(dim)
        self.feat_norm = norm_layer(dim)
        self.attn = MSDeformAttn(d_model=dim, n_levels=n_levels, n_heads=num_heads,
                                 n_points=n_points, ratio=deform_ratio)
        self.gamma = nn.Parameter(init_values * torch.ones(dim), requires_grad=True)

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index, keep_feat=False,
                feat_normed=None):
        """keep_feat: see Extractor.forward (here ``feat`` is c, which the extractor consumes next).
        feat_normed: ``self.feat_norm(feat)`` when the caller already has it."""
        def body(query, feat, fn=None):
            query, qn = fused.layer_norm_keep(self.query_norm, query)
            if fn is None:
                feat, fn = fused.layer_norm_keep(self.feat_norm, feat, fan_out=True)
            attn = self.attn(qn, reference_points, fn, spatial_shapes, level_start_index, None)
            return fused.residual(query, attn, self.gamma), feat

        if self.with_cp and query.requires_grad:
            out, feat = cp.checkpoint(body, query, feat, feat_normed, use_reentrant=False)
        else:
            out, feat = body(query, feat, feat_normed)
        return (out, feat) if keep_feat else out


class InteractionBlock(nn.Module):
    """injector -> a run of ViT blocks -> extractor (+ two extra extractors on the last block)."""
    with_cls = False

    def __init__(self, dim, num_heads=6, n_points=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 drop=0., drop_path=0., with_cffn=True, cffn_ratio=0.25, init_values=0.,
                 deform_ratio=1.0, extra_extractor=False, with_cp=False):
        super().__init__()
        self.injector = Injector(dim=dim, n_levels=3, num_heads=num_heads,
                                 init_values=init_values, n_points=n_points,
                                 norm_layer=norm_layer, deform_ratio=deform_ratio,
                                 with_cp=with_cp)
        ext = dict(dim=dim, num_heads=num_heads, n_points=n_points, norm_layer=norm_layer,
                   deform_ratio=deform_ratio, with_cffn=with_cffn, cffn_ratio=cffn_ratio,
                   drop=drop, drop_path=drop_path, with_cp=with_cp)
        self.extractor = Extractor(n_levels=1, **ext)
        if extra_extractor:
            self.extra_extractors = nn.Sequential(*[Extractor(**ext) for _ in range(2)])
        else:
            self.extra_extractors = None

    def _extract(self, x, c, deform_inputs2, H, W, c_normed=None):
        stages = [self.extractor] + (list(self.extra_extractors) if self.extra_extractors is not None else [])
        for k, stage in enumerate(stages):
            c, x = stage(query=c, reference_points=deform_inputs2[0], feat=x,
                         spatial_shapes=deform_inputs2[1], level_start_index=deform_inputs2[2],
                         H=H, W=W, keep_feat=True, query_normed=c_normed if k == 0 else None)
        return x, c

    def _inject(self, x, c, deform_inputs1):
        """x <- injector(x, c).  c is normalised here for BOTH its consumers - the injector's feat_norm
        and, since the injector leaves c unchanged, the first extractor's query_norm - with shared
        statistics (one read of the 132 MB c stream, one backward pass).  Returns (x, c, query_norm(c))."""
        c, cn_inj, cn_ext = fused.layer_norm_dual_keep(self.injector.feat_norm, self.extractor.query_norm, c)
        x = self.injector(query=x, reference_points=deform_inputs1[0], feat=c,
                          spatial_shapes=deform_inputs1[1], level_start_index=deform_inputs1[2],
                          feat_normed=cn_inj)
        return x, c, cn_ext

    def forward(self, x, c, blocks, deform_inputs1, deform_inputs2, H, W):
        x, c, cn = self._inject(x, c, deform_inputs1)
        x = run_blocks(blocks, x, H, W)
        return self._extract(x, c, deform_inputs2, H, W, cn)


class InteractionBlockWithCls(InteractionBlock):
    """Same, for trunks that carry a class token through the ViT blocks (BEiT adapter)."""
    with_cls = True

    def forward(self, x, c, cls, blocks, deform_inputs1, deform_inputs2, H, W):
        x, c, cn = self._inject(x, c, deform_inputs1)
        x = torch.cat((cls, x), dim=1)
        x = run_blocks(blocks, x, H, W)
        cls, x = x[:, :1], x[:, 1:]
        x, c = self._extract(x, c, deform_inputs2, H, W, cn)
        return x, c, cls


class SpatialPriorModule(nn.Module):
    """Convolutional stem producing the 1/4, 1/8, 1/16, 1/32 prior maps, projected to embed_dim.
    c1 stays a map (B, E, H/4, W/4); c2..c4 are returned as token sequences."""

    def __init__(self, inplanes=64, embed_dim=384, with_cp=False):
        super().__init__()
        self.with_cp = with_cp

        def conv_bn_relu(cin, cout, stride):
            return [nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False),
                    nn.SyncBatchNorm(cout), nn.ReLU(inplace=True)]

        self.stem = nn.Sequential(*(conv_bn_relu(3, inplanes, 2) + conv_bn_relu(inplanes, inplanes, 1) +
                                    conv_bn_relu(inplanes, inplanes, 1) +
                                    [nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]))
        self.conv2 = nn.Sequential(*conv_bn_relu(inplanes, 2 * inplanes, 2))
        self.conv3 = nn.Sequential(*conv_bn_relu(2 * inplanes, 4 * inplanes, 2))
        self.conv4 = nn.Sequential(*conv_bn_relu(4 * inplanes, 4 * inplanes, 2))
        self.fc1 = nn.Conv2d(inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)
        self.fc2 = nn.Conv2d(2 * inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)
        self.fc3 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)
        self.fc4 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1, stride=1, padding=0, bias=True)

    @staticmethod
    def _run(seq, x):
        """``seq(x)`` with every (BatchNorm, ReLU) pair of the Sequential as one fused op."""
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.modules.batchnorm._BatchNorm) and i + 1 < len(mods)
                    and isinstance(mods[i + 1], nn.ReLU)):
                x = fused.bn_relu(m, x)
                i += 2
            elif isinstance(m, nn.MaxPool2d):
                x = fused.max_pool(m, x)
                i += 1
            else:
                x = m(x)
                i += 1
        return x

    def _body(self, x):
        c1 = self._run(self.stem, x)
        c2 = self._run(self.conv2, c1)
        c3 = self._run(self.conv3, c2)
        c4 = self._run(self.conv4, c3)
        # bias_free_c1: the 1x1 conv to embed_dim without its bias - the caller folds fc1.bias into the
        # BatchNorm tail (fused.bn_tail shift) instead of a 100 M-element bias-add pass
        c1 = fused.conv1x1(self.fc1, c1) if self._bias_free_c1 else self.fc1(c1)
        if self._raw_maps:
            # the 1x1 convs without bias, still as maps: the caller adds bias + level embedding while
            # it lays the tokens out (fused.maps_to_tokens)
            return (c1, *(fused.conv1x1(f, c) for f, c in ((self.fc2, c2), (self.fc3, c3), (self.fc4, c4))))
        tokens = [f(c).flatten(2).transpose(1, 2) for f, c in
                  ((self.fc2, c2), (self.fc3, c3), (self.fc4, c4))]
        return (c1, *tokens)

    _bias_free_c1 = False
    _raw_maps = False

    def forward(self, x, bias_free_c1=False, raw_maps=False):
        self._bias_free_c1, self._raw_maps = bool(bias_free_c1), bool(raw_maps)
        try:
            if self.with_cp and x.requires_grad:
                return cp.checkpoint(self._body, x, use_reentrant=False)
            return self._body(x)
        finally:
            self._bias_free_c1 = self._raw_maps = False
