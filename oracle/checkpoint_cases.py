"""oracle/checkpoint_cases.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Seeded synthetic checkpoints for the pretrained-weight loader: the same dicts are written to disk and fed to the
reference's load_checkpoint by tools/gen_golden_checkpoint.py (expected tensors -> tests/golden/checkpoint.npz) and
to vitadapter/checkpoint.py by tests/test_checkpoint.py.  Nothing here reads /root/reference.
"""
from . import seeded

_SMALL = dict(patch_size=16, embed_dim=32, depth=2, num_heads=2, mlp_ratio=2, qkv_bias=True, init_values=1e-6,
              drop_path_rate=0.)

CASES = {
    # absolute position embedding of a 4x4-patch checkpoint (+ class token) into a 6x6-patch model: bicubic resize
    'pos_embed_resize': dict(
        model=dict(img_size=96, use_abs_pos_emb=True, use_rel_pos_bias=False, **_SMALL),
        check=['pos_embed', 'blocks.0.attn.qkv.weight', 'blocks.1.mlp.fc2.bias']),
    # one shared relative-position-bias table expanded to every block; relative_position_index buffers dropped;
    # `state_dict` wrapper and `module.` prefixes
    'shared_rel_pos_bias': dict(
        model=dict(img_size=64, use_abs_pos_emb=False, use_rel_pos_bias=True, **_SMALL),
        check=['blocks.0.attn.relative_position_bias_table', 'blocks.1.attn.relative_position_bias_table',
               'blocks.0.attn.relative_position_index', 'blocks.1.attn.proj.weight']),
}


# detection flavour (detection/mmcv_custom/checkpoint.py:379-445): per-block tables WITHOUT class-token rows in the model
# ((2 w - 1)^2 rows for a window of w), 3 more rows in the checkpoint that the loader always drops.  Equal-size branch
# only (the other one needs scipy's interp2d, gone from this image): 4 x 4 token grid, windows of 2 and the global 4.
DET_CASES = {
    'det_tables_equal_size': dict(
        model=dict(img_size=64, use_abs_pos_emb=False, use_rel_pos_bias=True, window_attn=[True, False],
                   window_size=[2, 4], **_SMALL),
        check=['blocks.0.attn.relative_position_bias_table', 'blocks.1.attn.relative_position_bias_table',
               'blocks.1.attn.proj.weight']),
}


def checkpoint(name):
    """The dict torch.save()d for case ``name``."""
    C = _SMALL['embed_dim']
    if name == 'det_tables_equal_size':
        sd = {
            'blocks.0.attn.relative_position_bias_table': seeded.randn('ckpt/det/t0', (3 * 3 + 3, _SMALL['num_heads']), 33),
            'blocks.1.attn.relative_position_bias_table': seeded.randn('ckpt/det/t1', (7 * 7 + 3, _SMALL['num_heads']), 33),
            'blocks.1.attn.proj.weight': seeded.randn('ckpt/det/proj', (C, C), 33),
        }
        return {'model': sd}
    if name == 'pos_embed_resize':
        sd = {
            'pos_embed': seeded.randn('ckpt/pos_embed', (1, 4 * 4 + 1, C), 31),
            'blocks.0.attn.qkv.weight': seeded.randn('ckpt/qkv', (3 * C, C), 31),
            'blocks.1.mlp.fc2.bias': seeded.randn('ckpt/fc2b', (C,), 31),
        }
        return sd
    if name == 'shared_rel_pos_bias':
        n = (2 * 4 - 1) * (2 * 4 - 1) + 3                 # 4x4 patches at 64 px
        sd = {
            'module.rel_pos_bias.relative_position_bias_table': seeded.randn('ckpt/rpb', (n, _SMALL['num_heads']), 32),
            'module.rel_pos_bias.relative_position_index': seeded.randn('ckpt/rpi', (17, 17), 32).long(),
            'module.blocks.1.attn.proj.weight': seeded.randn('ckpt/proj', (C, C), 32),
        }
        return {'state_dict': sd}
    raise KeyError(name)
