"""Minimal pretrained-weight loader for the ViT trunk (the reference's my_load_checkpoint,
/root/reference/segmentation/mmcv_custom/my_checkpoint.py:86-141, minus URL schemes):
reads a local file with a loader that executes nothing from it, strips the usual wrapper keys
/ prefixes and loads non-strictly."""
import torch


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    if filename.endswith('.safetensors'):
        from safetensors.torch import load_file
        ckpt = load_file(filename, device=str(map_location))
    else:
        ckpt = torch.load(filename, map_location=map_location, weights_only=True)
    if not isinstance(ckpt, dict):
        raise RuntimeError('No state_dict found in checkpoint file %s' % filename)
    for key in ('state_dict', 'model', 'module'):
        if key in ckpt and isinstance(ckpt[key], dict):
            ckpt = ckpt[key]
    state = {}
    for k, v in ckpt.items():
        for prefix in ('module.', 'backbone.'):
            if k.startswith(prefix):
                k = k[len(prefix):]
        state[k] = v
    missing, unexpected = model.load_state_dict(state, strict=strict)
    if logger is not None:
        logger.info('load_checkpoint: missing=%s unexpected=%s', missing, unexpected)
    return ckpt
