"""vitadapter.layer_decay against the reference's LayerDecayOptimizerConstructor: golden produced by
tools/gen_golden_layer_decay.py from the reference's own add_params (group names, order, member
parameters, lr, lr scale, weight decay)."""
import json
import os

import torch


def _tree():
    from vitadapter.backbones.vit_adapter import build_preset
    root = torch.nn.Module()
    root.backbone = build_preset('tiny_seg')
    head = torch.nn.Module()
    head.query_embed = torch.nn.Embedding(4, 8)
    head.query_feat = torch.nn.Embedding(4, 8)
    head.level_embed = torch.nn.Embedding(3, 8)
    head.cls_embed = torch.nn.Linear(8, 5)
    head.mask_embed = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 8))
    head.conv_seg = torch.nn.Conv2d(8, 3, 1)
    root.decode_head = head
    return root


def test_param_groups_match_reference(golden_dir):
    from vitadapter import layer_decay
    gold = json.load(open(os.path.join(golden_dir, 'layer_decay.json')))
    tree = _tree()
    for key, want in gold.items():
        num_layers, rate = key.split('_')
        got = layer_decay.param_groups(tree, 6e-5, 0.01, int(num_layers), float(rate))
        assert [g['group_name'] for g in got] == [w['group_name'] for w in want]
        for g, w in zip(got, want):
            assert g['param_names'] == w['param_names'], g['group_name']
            assert abs(g['lr'] - w['lr']) <= 1e-12 and abs(g['lr_scale'] - w['lr_scale']) <= 1e-12
            assert g['weight_decay'] == w['weight_decay']
            assert len(g['params']) == len(g['param_names'])
    opt = torch.optim.AdamW(layer_decay.param_groups(tree, 6e-5, 0.01, 12, 0.95))       # usable as is
    assert len(opt.param_groups) == len(gold['12_0.95'])


def test_layer_id_rules():
    from vitadapter.layer_decay import layer_id
    assert layer_id('backbone.pos_embed', 14) == 0 and layer_id('backbone.patch_embed.proj.weight', 14) == 0
    assert layer_id('backbone.blocks.0.attn.qkv.weight', 14) == 1 and layer_id('backbone.layers.11.x', 14) == 12
    assert layer_id('backbone.interactions.0.injector.gamma', 14) == 13
    assert layer_id('decode_head.query_embed.weight', 14) == 0 and layer_id('decode_head.conv_seg.weight', 14) == 13
