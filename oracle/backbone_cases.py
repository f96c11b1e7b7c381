"""oracle/backbone_cases.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Config literals + seeded inputs of the backbone goldens, shared by tools/gen_golden_backbone.py
(feeds the reference's classes) and the tests (feed the oracle restatement and the HIP-backed
product modules).  Small on purpose; D = 32 per deformable head as in every reference config.
"""
import torch

from . import cases, seeded

_SMALL = dict(patch_size=16, embed_dim=64, depth=4, num_heads=1, mlp_ratio=4, drop_path_rate=0.,
              conv_inplane=16, n_points=4, deform_num_heads=2, cffn_ratio=0.25, deform_ratio=1.0,
              interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]])

FULL_CASES = {
    # seg flavour, all-global attention (configs[1] family), square input
    'seg_glob_64': dict(cfg=dict(flavour='seg', window_attn=[False] * 4, window_size=[None] * 4,
                                 **_SMALL), hw=(64, 64), batch=2, modes=('eval', 'train')),
    # det flavour, window/global mix (configs[2] family), non-square input -> padded windows
    'det_win_96x128': dict(cfg=dict(flavour='det', window_attn=[True, False, True, False],
                                    window_size=[14, None, 14, None],
                                    **dict(_SMALL, deform_ratio=0.5, deform_num_heads=1)),
                           hw=(96, 128), batch=1, modes=('eval', 'train')),
    # det flavour with windows, no extra extractors, no ConvFFN, 2 blocks per interaction, only 2
    # interactions (the seg forward needs exactly 4: S/vit_adapter.py:126 unpacks four stage maps)
    'det_win_64x96': dict(cfg=dict(flavour='det', window_attn=[True, True, False, False],
                                   window_size=[14, 14, None, None], use_extra_extractor=False,
                                   with_cffn=False,
                                   **dict(_SMALL, interaction_indexes=[[0, 1], [2, 3]])),
                          hw=(64, 96), batch=2, modes=('eval',)),
}


def full_input(name):
    c = FULL_CASES[name]
    return seeded.randn('full/%s/x' % name, (c['batch'], 3) + tuple(c['hw']), 11)


def full_gouts(name, shapes):
    return [seeded.randn('full/%s/g%d' % (name, k), s, 11) for k, s in enumerate(shapes)]


PART = dict(embed=64, deform_heads=2, ratio=1.0, heads=1, tokens=4, inplanes=16, batch=2)


def part_geometry():
    """(injector geometry, extractor geometry) for a 64x64 image: [ref points, shapes, lsi]."""
    t = PART['tokens']
    pyramid = [(2 * t, 2 * t), (t, t), (t // 2, t // 2)]
    vit = [(t, t)]
    out = []
    for vshapes, qshapes in ((pyramid, vit), (vit, pyramid)):
        out.append([cases.reference_grid(qshapes), torch.as_tensor(vshapes, dtype=torch.long),
                    cases.level_start_index(vshapes)])
    return out


def part_tokens():
    t, E, B = PART['tokens'], PART['embed'], PART['batch']
    x = seeded.randn('part/x', (B, t * t, E), 12)
    c = seeded.randn('part/c', (B, 21 * (t // 2) ** 2, E), 12)
    return x, c


def part_image():
    t = PART['tokens']
    return seeded.randn('part/img', (PART['batch'], 3, 16 * t, 16 * t), 12)


def part_gout(tag, shape):
    return seeded.randn('part/g/' + tag, tuple(shape), 12)


# name: (windowed, H, W) token grids; window 14 -> 10x17 pads to 14x28 (two windows)
BLOCK_CASES = {'blk_global': (False, 6, 7), 'blk_window': (True, 10, 17), 'blk_window_exact': (True, 14, 14)}


def block_tokens(name):
    _, H, W = BLOCK_CASES[name]
    return seeded.randn('part/blk/' + name, (PART['batch'], H * W, PART['embed']), 12)


# ---- BEiT adapter: fixed input size = img_size (the bias table is built for its grid)
BEIT_CASES = {
    'beit_seg_64': dict(cfg=dict(img_size=64, patch_size=16, embed_dim=64, depth=4, num_heads=2, mlp_ratio=4,
                                 qkv_bias=True, use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=1e-6,
                                 drop_path_rate=0., conv_inplane=16, n_points=4, deform_num_heads=2,
                                 cffn_ratio=0.25, deform_ratio=1.0, with_cp=False,
                                 interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]]),
                        hw=(64, 64), batch=2, modes=('eval', 'train')),
    # two blocks per interaction (class token carried across blocks), deform_ratio 0.5
    'beit_seg_96': dict(cfg=dict(img_size=96, patch_size=16, embed_dim=64, depth=4, num_heads=1, mlp_ratio=2,
                                 qkv_bias=True, use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=1e-6,
                                 drop_path_rate=0., conv_inplane=16, n_points=4, deform_num_heads=1,
                                 cffn_ratio=0.25, deform_ratio=0.5, with_cp=False,
                                 interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]]),
                        hw=(96, 96), batch=1, modes=('train',)),
}


# detection flavour (mmdet_custom/models/backbones/beit_adapter.py): no class token, windowed blocks (grid padded
# BEFORE the projection) and global blocks whose window is the whole grid, per-block (2w-1)^2 bias tables
BEIT_DET_CASES = {
    # 4 x 4 token grid: windows of 2 (no padding), global blocks with window 4
    'beit_det_64': dict(cfg=dict(img_size=64, patch_size=16, embed_dim=64, depth=4, num_heads=2, mlp_ratio=4,
                                 qkv_bias=True, use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=1e-6,
                                 drop_path_rate=0., conv_inplane=16, n_points=4, deform_num_heads=2,
                                 cffn_ratio=0.25, deform_ratio=1.0, with_cp=False, version='new',
                                 window_attn=[True, False, True, False], window_size=[2, 4, 2, 4],
                                 interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]]),
                        hw=(64, 64), batch=2, modes=('eval', 'train')),
    # 6 x 6 token grid: windows of 4 (padded to 8 x 8: padded tokens carry q_bias / v_bias), global window 6,
    # absolute position embedding resized from a 2 x 2 pretraining grid, version 'old' (a map per interaction)
    'beit_det_96': dict(cfg=dict(img_size=96, patch_size=16, embed_dim=64, depth=4, num_heads=1, mlp_ratio=2,
                                 qkv_bias=True, use_abs_pos_emb=False, use_rel_pos_bias=True, init_values=1e-6,
                                 drop_path_rate=0., conv_inplane=16, n_points=4, deform_num_heads=1,
                                 cffn_ratio=0.25, deform_ratio=0.5, with_cp=False, version='old',
                                 window_attn=[True, True, False, True], window_size=[4, 4, 6, 4],
                                 interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]]),
                        hw=(96, 96), batch=1, modes=('train',)),
}


def beit_det_input(name):
    c = BEIT_DET_CASES[name]
    return seeded.randn('beit_det/%s/x' % name, (c['batch'], 3) + tuple(c['hw']), 13)


def beit_det_gouts(name, shapes):
    return [seeded.randn('beit_det/%s/g%d' % (name, k), s, 13) for k, s in enumerate(shapes)]


def beit_input(name):
    c = BEIT_CASES[name]
    return seeded.randn('beit/%s/x' % name, (c['batch'], 3) + tuple(c['hw']), 13)


def beit_gouts(name, shapes):
    return [seeded.randn('beit/%s/g%d' % (name, k), s, 13) for k, s in enumerate(shapes)]


def float_shapes(module):
    """{key: shape} of the floating-point state_dict entries (+ num_batches_tracked): integer index
    buffers such as relative_position_index are structural and keep the module's own values."""
    return {k: tuple(v.shape) for k, v in module.state_dict().items()
            if v.is_floating_point() or k.endswith('num_batches_tracked')}


# ---- full-size headline workloads (SURVEY 8c G5 "checksums for 512^2 batch 2") -----------------
# BASELINE configs[1] (ViT-Adapter-T, 512 x 512, batch 2) and configs[2] (ViT-Adapter-B det flavour, 1024 x 1024, one
# image: the per-GPU batch of 2 is two independent images, SyncBN statistics aside).  Train mode, drop_path 0 (no RNG in
# the graph), seeded weights.  The fixture holds DIGESTS: per output the fp64 sum + 4096 sampled elements, per parameter
# gradient seeded.digest + its L2 norm (tools/gen_golden_fullsize.py, from the reference's own classes).
FULLSIZE_CASES = {
    'tiny_seg_512': dict(cfg=dict(flavour='seg', patch_size=16, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4,
                                  drop_path_rate=0., conv_inplane=64, n_points=4, deform_num_heads=6, cffn_ratio=0.25,
                                  deform_ratio=1.0, interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]],
                                  window_attn=[False] * 12, window_size=[None] * 12),
                         hw=(512, 512), batch=2),
    'base_det_1024': dict(cfg=dict(flavour='det', patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                                   drop_path_rate=0., conv_inplane=64, n_points=4, deform_num_heads=12, cffn_ratio=0.25,
                                   deform_ratio=0.5, interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]],
                                   window_attn=[True, True, False] * 4, window_size=[14, 14, None] * 4),
                          hw=(1024, 1024), batch=1),
}
FULLSIZE_SAMPLES = 4096


def fullsize_input(name):
    c = FULLSIZE_CASES[name]
    return seeded.randn('fullsize/%s/x' % name, (c['batch'], 3) + tuple(c['hw']), 21)


def fullsize_gouts(name, shapes):
    """Seeded output gradients, scaled so that every level contributes a gradient of similar size."""
    return [seeded.randn('fullsize/%s/g%d' % (name, k), s, 21) for k, s in enumerate(shapes)]


def fullsize_positions(key, numel):
    """FULLSIZE_SAMPLES seeded flat positions into a tensor of ``numel`` elements."""
    g = torch.Generator(device='cpu')
    import zlib
    g.manual_seed(zlib.crc32(('pos/' + key).encode()) % (2 ** 31 - 1))
    return torch.randint(0, numel, (FULLSIZE_SAMPLES,), generator=g)
