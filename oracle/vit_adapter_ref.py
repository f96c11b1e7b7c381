"""oracle/vit_adapter_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional CPU restatement of the reference's ViT-Adapter backbone forward, driven by a
state_dict with the reference's key names.  Plain torch ops only (autograd gives reference
gradients); deformable attention goes through ``oracle.msda.core_torch``.  Used by tests as the
checker for the HIP-backed ``vitadapter.backbones.ViTAdapter`` and by bench.py's
``cpu_baseline`` leg.  Nothing under vit-adapter_amd/ imports this file.

Each function cites the reference lines it restates (paths under /root/reference):
  S = segmentation/mmseg_custom/models/backbones,  D = detection/mmdet_custom/models/backbones,
  O = detection/ops/modules/ms_deform_attn.py.
Pinned by tests/test_backbone_oracle.py against tests/golden/backbone_*.npz, made by
tools/gen_golden_backbone.py from the reference's own classes.
"""
import math

import torch
import torch.nn.functional as F

from . import msda as _msda


class Cfg:
    """The constructor keywords that shape the forward (S/vit_adapter.py:21-28, base/vit.py:260-265)."""

    def __init__(self, flavour='seg', embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 pretrain_size=224, conv_inplane=64, n_points=4, deform_num_heads=6,
                 deform_ratio=1.0, cffn_ratio=0.25, with_cffn=True, interaction_indexes=None,
                 window_attn=False, window_size=14, layer_scale=True, add_vit_feature=True,
                 use_extra_extractor=True, patch_size=16, **ignored):
        self.flavour = flavour
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.pretrain_size = pretrain_size
        self.n_points = n_points
        self.deform_num_heads = deform_num_heads
        self.deform_ratio = deform_ratio
        self.with_cffn = with_cffn
        self.interaction_indexes = interaction_indexes
        self.window_attn = window_attn if isinstance(window_attn, (list, tuple)) else [window_attn] * depth
        self.window_size = window_size if isinstance(window_size, (list, tuple)) else [window_size] * depth
        self.layer_scale = layer_scale
        self.add_vit_feature = add_vit_feature
        self.use_extra_extractor = use_extra_extractor
        self.patch_size = patch_size


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _ln(x, sd, name, eps=1e-6):
    return F.layer_norm(x, x.shape[-1:], sd[name + '.weight'], sd[name + '.bias'], eps)


def _lin(x, sd, name):
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def _bn(x, sd, name, training, stats_out=None):
    """nn.SyncBatchNorm on one process == batch_norm (eval: running stats; train: batch stats)."""
    return F.batch_norm(x, None if training else sd[name + '.running_mean'],
                        None if training else sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'], training, 0.1, 1e-5)


def reference_points(shapes):
    """S/adapter_modules.py:13-25."""
    pts = []
    for h, w in shapes:
        ys = torch.linspace(0.5, h - 0.5, h) / h
        xs = torch.linspace(0.5, w - 0.5, w) / w
        gy, gx = torch.meshgrid(ys, xs, indexing='ij')
        pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1)[None])
    return torch.cat(pts, 1)[:, :, None]


def ms_deform_attn_module(sd, query, ref, feat, shapes, M, P, core=None):
    """O:83-130: value_proj -> offsets / softmax(weights) -> loc = ref + off/(W,H) -> core -> output_proj.
    ``shapes``: list of (H, W)."""
    core = core or _msda.core_torch
    N, Lq, _ = query.shape
    S = feat.shape[1]
    L = len(shapes)
    value = _lin(feat, sd, 'value_proj')
    value = value.view(N, S, M, value.shape[-1] // M)
    off = _lin(query, sd, 'sampling_offsets').view(N, Lq, M, L, P, 2)
    w = F.softmax(_lin(query, sd, 'attention_weights').view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    norm = torch.tensor([[wd, ht] for ht, wd in shapes], dtype=query.dtype)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = core(value, shapes, loc, w)
    return _lin(out, sd, 'output_proj')


def injector(sd, x, c, ref, shapes, cfg):
    """S/adapter_modules.py:138-152."""
    a = ms_deform_attn_module(_sub(sd, 'attn.'), _ln(x, sd, 'query_norm'), ref, _ln(c, sd, 'feat_norm'),
                              shapes, cfg.deform_num_heads, cfg.n_points)
    return x + sd['gamma'] * a


def dwconv(sd, x, H, W):
    """S/adapter_modules.py:72-87: shared depthwise 3x3 over the 16n | 4n | n token maps."""
    B, N, C = x.shape
    n = N // 21
    outs = []
    for lo, hi, (h, w) in ((0, 16 * n, (2 * H, 2 * W)), (16 * n, 20 * n, (H, W)), (20 * n, N, (H // 2, W // 2))):
        m = x[:, lo:hi].transpose(1, 2).reshape(B, C, h, w).contiguous()
        m = F.conv2d(m, sd['dwconv.weight'], sd['dwconv.bias'], 1, 1, 1, C)
        outs.append(m.flatten(2).transpose(1, 2))
    return torch.cat(outs, 1)


def extractor(sd, c, x, ref, shapes, H, W, cfg):
    """S/adapter_modules.py:106-124 (DropPath is the identity in the oracle: drop_path = 0)."""
    a = ms_deform_attn_module(_sub(sd, 'attn.'), _ln(c, sd, 'query_norm'), ref, _ln(x, sd, 'feat_norm'),
                              shapes, cfg.deform_num_heads, cfg.n_points)
    c = c + a
    if cfg.with_cffn:
        f = _sub(sd, 'ffn.')
        y = _lin(_ln(c, sd, 'ffn_norm'), f, 'fc1')
        y = F.gelu(dwconv(_sub(f, 'dwconv.'), y, H, W))
        c = c + _lin(y, f, 'fc2')
    return c


def global_attention(sd, x, heads):
    """D/base/vit.py:78-91."""
    B, N, C = x.shape
    qkv = _lin(x, sd, 'qkv').reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(-1)
    return _lin((a @ v).transpose(1, 2).reshape(B, N, C), sd, 'proj')


def windowed_attention(sd, x, H, W, heads, ws):
    """D/base/vit.py:136-167: project, THEN zero-pad to a multiple of ws, attend per window
    (padded tokens included, unmasked), crop."""
    B, N, C = x.shape
    Hp, Wp = math.ceil(H / ws) * ws, math.ceil(W / ws) * ws
    qkv = _lin(x, sd, 'qkv').transpose(1, 2).reshape(B, 3 * C, H, W)
    qkv = F.pad(qkv, [0, Wp - W, 0, Hp - H])
    qkv = F.unfold(qkv, kernel_size=(ws, ws), stride=(ws, ws))
    nwin = qkv.shape[-1]
    qkv = qkv.reshape(B, 3 * C, ws * ws, nwin).permute(0, 3, 2, 1)
    qkv = qkv.reshape(B, nwin, ws * ws, 3, heads, C // heads).permute(3, 0, 1, 4, 2, 5)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(-1)
    y = (a @ v).permute(0, 2, 4, 3, 1).reshape(B, C * ws * ws, nwin)
    y = F.fold(y, output_size=(Hp, Wp), kernel_size=(ws, ws), stride=(ws, ws))
    y = y[:, :, :H, :W].reshape(B, C, N).transpose(-1, -2)
    return _lin(y, sd, 'proj')


def block(sd, x, H, W, heads, windowed, ws, layer_scale):
    """D/base/vit.py:232-248 (drop_path = 0, no residual conv branch)."""
    y = _ln(x, sd, 'norm1')
    a = windowed_attention(_sub(sd, 'attn.'), y, H, W, heads, ws) if windowed else \
        global_attention(_sub(sd, 'attn.'), y, heads)
    x = x + (sd['gamma1'] * a if layer_scale else a)
    y = _ln(x, sd, 'norm2')
    f = _lin(F.gelu(_lin(y, sd, 'mlp.fc1')), sd, 'mlp.fc2')
    return x + (sd['gamma2'] * f if layer_scale else f)


def spm(sd, x, training):
    """S/adapter_modules.py:272-296."""
    def cbr(x, conv, bn, stride):
        return F.relu(_bn(F.conv2d(x, sd[conv + '.weight'], None, stride, 1), sd, bn, training))
    c1 = cbr(x, 'stem.0', 'stem.1', 2)
    c1 = cbr(c1, 'stem.3', 'stem.4', 1)
    c1 = cbr(c1, 'stem.6', 'stem.7', 1)
    c1 = F.max_pool2d(c1, 3, 2, 1)
    c2 = cbr(c1, 'conv2.0', 'conv2.1', 2)
    c3 = cbr(c2, 'conv3.0', 'conv3.1', 2)
    c4 = cbr(c3, 'conv4.0', 'conv4.1', 2)
    proj = [F.conv2d(c, sd[f + '.weight'], sd[f + '.bias']) for f, c in
            (('fc1', c1), ('fc2', c2), ('fc3', c3), ('fc4', c4))]
    return (proj[0],) + tuple(p.flatten(2).transpose(1, 2) for p in proj[1:])


def interaction(sd, i, x, c, full_sd, cfg, H, W, geo1, geo2):
    """S/adapter_modules.py:177-191."""
    x = injector(_sub(sd, 'injector.'), x, c, geo1[0], geo1[1], cfg)
    lo, hi = cfg.interaction_indexes[i][0], cfg.interaction_indexes[i][-1]
    for b in range(lo, hi + 1):
        x = block(_sub(full_sd, 'blocks.%d.' % b), x, H, W, cfg.num_heads, cfg.window_attn[b],
                  cfg.window_size[b], cfg.layer_scale)
    c = extractor(_sub(sd, 'extractor.'), c, x, geo2[0], geo2[1], H, W, cfg)
    if any(k.startswith('extra_extractors.') for k in sd):
        for j in range(2):
            c = extractor(_sub(sd, 'extra_extractors.%d.' % j), c, x, geo2[0], geo2[1], H, W, cfg)
    return x, c


def vit_adapter_forward(sd, x, cfg, training=False):
    """S/vit_adapter.py:93-137 (seg) and D/vit_adapter.py:91-132 (det)."""
    _, _, h, w = x.shape
    pyramid = [(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)]
    vit = [(h // 16, w // 16)]
    geo1 = (reference_points(vit), pyramid)        # injector: ViT tokens query the SPM pyramid
    geo2 = (reference_points(pyramid), vit)        # extractor: SPM tokens query the ViT map

    c1, c2, c3, c4 = spm(_sub(sd, 'spm.'), x, training)
    c2, c3, c4 = c2 + sd['level_embed'][0], c3 + sd['level_embed'][1], c4 + sd['level_embed'][2]
    n2, n3 = c2.shape[1], c3.shape[1]
    c = torch.cat([c2, c3, c4], 1)

    t = F.conv2d(x, sd['patch_embed.proj.weight'], sd.get('patch_embed.proj.bias'), cfg.patch_size)
    bs, dim, H, W = t.shape
    t = t.flatten(2).transpose(1, 2)
    ps = cfg.pretrain_size // 16
    pe = sd['pos_embed'][:, 1:].reshape(1, ps, ps, -1).permute(0, 3, 1, 2)
    pe = F.interpolate(pe, size=(H, W), mode='bicubic', align_corners=False)
    t = t + pe.reshape(1, -1, H * W).permute(0, 2, 1)

    stage = []
    for i in range(len(cfg.interaction_indexes)):
        t, c = interaction(_sub(sd, 'interactions.%d.' % i), i, t, c, sd, cfg, H, W, geo1, geo2)
        stage.append(t.transpose(1, 2).reshape(bs, dim, H, W).contiguous())

    c2 = c[:, :n2].transpose(1, 2).reshape(bs, dim, 2 * H, 2 * W).contiguous()
    c3 = c[:, n2:n2 + n3].transpose(1, 2).reshape(bs, dim, H, W).contiguous()
    c4 = c[:, n2 + n3:].transpose(1, 2).reshape(bs, dim, H // 2, W // 2).contiguous()
    c1 = F.conv_transpose2d(c2, sd['up.weight'], sd['up.bias'], 2) + c1
    if cfg.add_vit_feature:
        x1, x2, x3, x4 = stage if cfg.flavour == 'seg' else [stage[-1]] * 4
        c1 = c1 + F.interpolate(x1, scale_factor=4, mode='bilinear', align_corners=False)
        c2 = c2 + F.interpolate(x2, scale_factor=2, mode='bilinear', align_corners=False)
        c3 = c3 + x3
        c4 = c4 + F.interpolate(x4, scale_factor=0.5, mode='bilinear', align_corners=False)
    return [_bn(f, sd, 'norm%d' % (k + 1), training) for k, f in enumerate((c1, c2, c3, c4))]
