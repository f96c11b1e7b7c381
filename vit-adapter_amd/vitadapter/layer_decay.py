"""Layer-wise learning-rate decay parameter groups.

What the reference's ``LayerDecayOptimizerConstructor`` computes
(ref: segmentation/mmcv_custom/layer_decay_optimizer_constructor.py:17-113), as plain functions
that return ``torch.optim`` parameter groups - no mmcv runner needed:

* depth index of a parameter (``layer_id``): embeddings (cls / mask / pos / visual / patch embed) and
  the Mask2Former query / level / class / mask embeddings of the decode head are depth 0; a ViT /
  BEiT block ``backbone.blocks.<i>.*`` (or ``backbone.layers.<i>.*``) is depth i + 1; everything
  else (adapter modules, necks, heads) takes the last depth ``num_max_layer - 1``;
* learning-rate scale ``decay ** (num_layers + 2 - depth - 1)``, with ``num_layers`` the config's
  ``paramwise_cfg.num_layers`` (the constructor itself adds 2);
* no weight decay for 1-D parameters, biases and the bare names pos_embed / cls_token /
  visual_embed.
"""
import re

_DEPTH0_EXACT = frozenset('backbone.' + n for n in ('cls_token', 'mask_token', 'pos_embed', 'visual_embed'))
_DEPTH0_PREFIX = ('backbone.patch_embed', 'backbone.visual_embed') + tuple(
    'decode_head.' + n for n in ('mask_embed', 'cls_embed', 'level_embed', 'query_embed', 'query_feat'))
_BLOCK = re.compile(r'^backbone\.(?:blocks|layers)[^.]*\.(\d+)')
_NO_DECAY_NAMES = frozenset(('pos_embed', 'cls_token', 'visual_embed'))


def layer_id(name, num_max_layer):
    """Depth index of parameter ``name`` (as seen from the segmentor / detector root)."""
    if name in _DEPTH0_EXACT or name.startswith(_DEPTH0_PREFIX):
        return 0
    if name.startswith(('backbone.blocks', 'backbone.layers')):
        return int(name.split('.')[2]) + 1
    return num_max_layer - 1


def param_groups(module, base_lr, weight_decay, num_layers, layer_decay_rate):
    """Parameter groups ``layer_<depth>_<decay|no_decay>`` with ``lr = base_lr * scale`` and the
    reference's extra keys (``lr_scale``, ``group_name``, ``param_names``), in first-seen order."""
    depth_count = num_layers + 2
    groups = {}
    for name, p in module.named_parameters():
        if not p.requires_grad:
            continue
        no_decay = p.dim() == 1 or name.endswith('.bias') or name in _NO_DECAY_NAMES
        depth = layer_id(name, depth_count)
        key = 'layer_%d_%s' % (depth, 'no_decay' if no_decay else 'decay')
        g = groups.get(key)
        if g is None:
            scale = layer_decay_rate ** (depth_count - depth - 1)
            g = groups[key] = dict(weight_decay=0. if no_decay else weight_decay, params=[], param_names=[],
                                   lr_scale=scale, group_name=key, lr=scale * base_lr)
        g['params'].append(p)
        g['param_names'].append(name)
    return list(groups.values())
