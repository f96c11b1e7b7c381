"""oracle/seeded.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Portable seeded tensors so that fixtures need to hold only expected OUTPUTS: the
golden generator (tools/gen_golden.py, run where /root/reference exists) and the tests
(run anywhere, including the GPU box where the reference is absent) regenerate
bit-identical inputs and weights from (key, shape, seed) with torch's CPU mt19937
generator.  A digest of every regenerated tensor is stored in the fixture and checked by
the tests, so a torch version whose CPU generator differs is detected instead of
silently comparing different inputs.
"""
import zlib

import numpy as np
import torch


def _gen(key, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * int(seed)) % (2 ** 31 - 1))
    return g


def randn(key, shape, seed=0, dtype=torch.float32):
    return torch.randn(tuple(shape), generator=_gen(key, seed), dtype=torch.float32).to(dtype)


def rand(key, shape, seed=0, dtype=torch.float32):
    return torch.rand(tuple(shape), generator=_gen(key, seed), dtype=torch.float32).to(dtype)


def digest(t):
    """Order-sensitive fp64 digest of a tensor (sum, weighted sum)."""
    a = t.detach().to(torch.float64).flatten().cpu().numpy()
    w = np.cos(np.arange(a.size, dtype=np.float64) * 0.61803398875)
    return np.array([a.sum(), (a * w).sum()], dtype=np.float64)


def seeded_param(key, shape, seed=0):
    """Deterministic non-degenerate value for a state_dict entry, chosen by its name.

    Deliberately NOT the reference's init: its zero-initialised injector ``gamma`` and
    ``sampling_offsets.weight`` would make whole branches no-ops and the parity tests blind.
    """
    shape = tuple(shape)
    if key.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.int64)
    if key.endswith('running_var'):
        return 0.5 + rand(key, shape, seed)
    if key.endswith('running_mean'):
        return 0.1 * randn(key, shape, seed)
    leaf = key.rsplit('.', 1)[-1]
    if leaf.startswith('gamma'):
        return 0.5 + 0.3 * randn(key, shape, seed)
    if leaf in ('pos_embed', 'level_embed', 'cls_token'):
        return 0.2 * randn(key, shape, seed)
    if leaf == 'bias':
        if 'sampling_offsets' in key:
            return 1.5 * randn(key, shape, seed)      # offsets are in pixels
        return 0.05 * randn(key, shape, seed)
    if leaf == 'weight' and len(shape) == 1:          # LayerNorm / BatchNorm scale
        return 1.0 + 0.1 * randn(key, shape, seed)
    if leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        if 'dwconv' in key:
            fan_in = int(np.prod(shape[2:]))
        if key.endswith('up.weight'):                 # ConvTranspose2d: (in, out, kh, kw)
            fan_in = shape[0]
        scale = 1.0 / np.sqrt(max(fan_in, 1))
        if 'sampling_offsets' in key:
            scale *= 0.5
        return scale * randn(key, shape, seed)
    return 0.1 * randn(key, shape, seed)


def seeded_state_dict(shapes, seed=0):
    """shapes: {key: shape} -> {key: tensor}."""
    return {k: seeded_param(k, s, seed) for k, s in shapes.items()}
