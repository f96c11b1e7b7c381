"""oracle/cases.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Seeded input builders shared by tools/gen_golden.py (which feeds them to the reference's
Python) and by the tests / smoke / bench (which feed the same tensors to the HIP path and
to the oracle).  Nothing here reads /root/reference.
"""
import math

import numpy as np
import torch

from . import seeded


def level_start_index(shapes):
    hw = torch.as_tensor(shapes, dtype=torch.long)
    return torch.cat((hw.new_zeros((1,)), hw.prod(1).cumsum(0)[:-1]))


def reference_grid(shapes):
    """Pixel-centre reference points (1, sum(H*W), 1, 2) as (x, y) in (0,1).

    Same values as the reference's get_reference_points
    (segmentation/mmseg_custom/models/backbones/adapter_modules.py:13-25)."""
    pts = []
    for h, w in shapes:
        ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
        xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
        gy, gx = torch.meshgrid(ys, xs, indexing='ij')
        pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    return torch.cat(pts, 0)[None, :, None, :]


def ring_offsets(M, L, P):
    """The reference's sampling-offset bias pattern, (M, L, P, 2)
    (detection/ops/modules/ms_deform_attn.py:64-72): unit ring direction per head x (p+1)."""
    th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    d = d.view(M, 1, 1, 2).repeat(1, L, P, 1)
    for p in range(P):
        d[:, :, p, :] *= p + 1
    return d


def msda_inputs(name, N, M, D, P, Lq, shapes, mode, seed=0, query_shapes=None,
                dtype=torch.float32):
    """Build (value, shapes, lsi, loc, attn, grad_out) for one MSDA call.

    mode: 'uniform'  loc ~ U[0,1)                      (detection/ops/test.py:29 recipe)
          'oob'      loc ~ U[-0.3,1.3)                 (exercises the zero-padding gates)
          'adapter'  loc = reference grid of ``query_shapes`` + (ring bias + N(0,1) px)/(W_l,H_l)
          'edge'     loc snapped onto exact pixel centres / borders (floor() boundary cases)
    """
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    hw = torch.as_tensor(shapes, dtype=torch.long)
    lsi = level_start_index(shapes)
    value = seeded.randn(name + '/value', (N, S, M, D), seed)
    attn = torch.softmax(seeded.randn(name + '/attn', (N, Lq, M, L * P), seed), -1).view(
        N, Lq, M, L, P)
    if mode == 'uniform':
        loc = seeded.rand(name + '/loc', (N, Lq, M, L, P, 2), seed)
    elif mode == 'oob':
        loc = seeded.rand(name + '/loc', (N, Lq, M, L, P, 2), seed) * 1.6 - 0.3
    elif mode == 'adapter':
        ref = reference_grid(query_shapes)                      # (1, Lq, 1, 2)
        assert ref.shape[1] == Lq
        off = ring_offsets(M, L, P)[None, None] + seeded.randn(
            name + '/off', (N, Lq, M, L, P, 2), seed)
        norm = torch.stack([hw[:, 1], hw[:, 0]], -1).to(torch.float32)   # (W_l, H_l)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    elif mode == 'edge':
        # locations whose pixel coordinate is an exact integer or exactly -1 / H: k/H + 0.5/H
        k = torch.floor(seeded.rand(name + '/k', (N, Lq, M, L, P, 2), seed) * 8.0) - 2.0
        norm = torch.stack([hw[:, 1], hw[:, 0]], -1).to(torch.float32)
        loc = (k + 0.5) / norm[None, None, None, :, None, :]
    else:
        raise ValueError(mode)
    gout = seeded.randn(name + '/gout', (N, Lq, M * D), seed)
    return (value.to(dtype), hw, lsi, loc.contiguous().to(dtype), attn.contiguous().to(dtype),
            gout.to(dtype))


# Golden cases: name -> kwargs for msda_inputs.  Kept SMALL on purpose: the fixture file
# holds the reference's outputs for these and pins the oracle; the oracle then checks the
# HIP path at any size (PARITY_CASES below) without further stored data.
ADAPTER_CASES = {}
for _mode in ('uniform', 'oob', 'adapter', 'edge'):
    ADAPTER_CASES['inj_' + _mode] = dict(
        N=2, M=6, D=32, P=4, Lq=16, shapes=[(8, 8), (4, 4), (2, 2)], mode=_mode,
        query_shapes=[(4, 4)])
    ADAPTER_CASES['ext_' + _mode] = dict(
        N=2, M=6, D=32, P=4, Lq=84, shapes=[(4, 4)], mode=_mode,
        query_shapes=[(8, 8), (4, 4), (2, 2)])
# non-square, non-power-of-two maps; D != 32 paths; many levels (config-1 shape family)
ADAPTER_CASES['rect_d32'] = dict(N=1, M=3, D=32, P=4, Lq=37, shapes=[(7, 11), (5, 3)],
                                 mode='oob')
ADAPTER_CASES['l4_d32'] = dict(N=1, M=8, D=32, P=4, Lq=21, shapes=[(8, 8), (4, 4), (2, 2), (1, 1)],
                               mode='oob')
ADAPTER_CASES['d16'] = dict(N=2, M=4, D=16, P=2, Lq=23, shapes=[(6, 6), (3, 3)], mode='oob')
ADAPTER_CASES['d64'] = dict(N=1, M=2, D=64, P=4, Lq=19, shapes=[(9, 5)], mode='oob')
ADAPTER_CASES['d24_p3'] = dict(N=1, M=5, D=24, P=3, Lq=31, shapes=[(5, 7), (4, 4), (3, 2)],
                               mode='oob')

# Oracle-checked parity cases (no stored outputs): the adapter's real call shapes scaled to
# sizes the scalar C oracle finishes in well under a second each.
PARITY_CASES = {}
for _mode in ('uniform', 'oob', 'adapter', 'edge'):
    PARITY_CASES['inj128_' + _mode] = dict(      # 128x128 image: T-config heads
        N=2, M=6, D=32, P=4, Lq=64, shapes=[(16, 16), (8, 8), (4, 4)], mode=_mode,
        query_shapes=[(8, 8)])
    PARITY_CASES['ext128_' + _mode] = dict(
        N=2, M=6, D=32, P=4, Lq=336, shapes=[(8, 8)], mode=_mode,
        query_shapes=[(16, 16), (8, 8), (4, 4)])
PARITY_CASES['inj256_b'] = dict(                 # 256x256 image, B-config heads (12 x 32)
    N=2, M=12, D=32, P=4, Lq=256, shapes=[(32, 32), (16, 16), (8, 8)], mode='adapter',
    query_shapes=[(16, 16)])
PARITY_CASES['ext256_b'] = dict(
    N=2, M=12, D=32, P=4, Lq=1344, shapes=[(16, 16)], mode='adapter',
    query_shapes=[(32, 32), (16, 16), (8, 8)])
PARITY_CASES['inj_rect_l'] = dict(               # 160x224 image, L-config heads (16 x 32)
    N=1, M=16, D=32, P=4, Lq=140, shapes=[(20, 28), (10, 14), (5, 7)], mode='adapter',
    query_shapes=[(10, 14)])
PARITY_CASES['ext_rect_l'] = dict(
    N=1, M=16, D=32, P=4, Lq=735, shapes=[(10, 14)], mode='adapter',
    query_shapes=[(20, 28), (10, 14), (5, 7)])
PARITY_CASES['cfg1_small'] = dict(               # config-1 family: 4 levels, 8 heads, Lq = S
    N=1, M=8, D=32, P=4, Lq=340, shapes=[(16, 16), (8, 8), (4, 4), (2, 2)], mode='uniform')
PARITY_CASES['pixdec_small'] = dict(             # Mask2Former pixel decoder family: M=8, L=3, Lq=S
    N=2, M=8, D=32, P=4, Lq=336, shapes=[(16, 16), (8, 8), (4, 4)], mode='adapter',
    query_shapes=[(16, 16), (8, 8), (4, 4)])
PARITY_CASES['ragged_q1'] = dict(N=1, M=1, D=32, P=1, Lq=1, shapes=[(1, 1)], mode='oob')
PARITY_CASES['ragged_q65'] = dict(N=3, M=5, D=32, P=4, Lq=65, shapes=[(3, 5)], mode='oob')
for _d in (1, 2, 4, 8, 16, 24, 30, 48, 64, 71, 128, 256):
    PARITY_CASES['generic_d%d' % _d] = dict(
        N=2, M=3, D=_d, P=3, Lq=29, shapes=[(6, 4), (3, 2)], mode='oob')


def testpy_inputs(D, dtype=torch.float64):
    """detection/ops/test.py:16-33 recipe: N,M=1,2; Lq,L,P=2,2,2; shapes [(6,4),(3,2)];
    value=rand*0.01, loc=rand, attn=(rand+1e-5) normalised over (L,P)."""
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = [(6, 4), (3, 2)]
    S = 30
    name = 'testpy/D%d' % D
    value = seeded.rand(name + '/value', (N, S, M, D), 3) * 0.01
    loc = seeded.rand(name + '/loc', (N, Lq, M, L, P, 2), 3)
    attn = seeded.rand(name + '/attn', (N, Lq, M, L, P), 3) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = seeded.randn(name + '/gout', (N, Lq, M * D), 3)
    hw = torch.as_tensor(shapes, dtype=torch.long)
    return (value.to(dtype), hw, level_start_index(shapes), loc.to(dtype), attn.to(dtype),
            gout.to(dtype))


TESTPY_CHANNELS = (30, 32, 64, 71, 1025, 2048, 3096)     # detection/ops/test.py:108


def bench_inputs(cfg, device='cpu', seed=0):
    """Full-size MSDA call shapes of the BASELINE configs.
    Generated with torch's generator on ``device`` (sizes here are too big for fixtures;
    full-size parity uses properties, not stored outputs)."""
    table = {
        # name: (N, M, D, P, Lq, shapes, query_shapes)
        'cfg1': (1, 8, 32, 4, 5440, [(64, 64), (32, 32), (16, 16), (8, 8)],
                 [(64, 64), (32, 32), (16, 16), (8, 8)]),
        'cfg2_inj': (2, 6, 32, 4, 1024, [(64, 64), (32, 32), (16, 16)], [(32, 32)]),
        'cfg2_ext': (2, 6, 32, 4, 5376, [(32, 32)], [(64, 64), (32, 32), (16, 16)]),
        'cfg3_inj': (2, 12, 32, 4, 4096, [(128, 128), (64, 64), (32, 32)], [(64, 64)]),
        'cfg3_ext': (2, 12, 32, 4, 21504, [(64, 64)], [(128, 128), (64, 64), (32, 32)]),
        'cfg4_inj': (2, 16, 32, 4, 1600, [(80, 80), (40, 40), (20, 20)], [(40, 40)]),
        'cfg4_ext': (2, 16, 32, 4, 8400, [(40, 40)], [(80, 80), (40, 40), (20, 20)]),
        'cfg5_inj': (1, 16, 32, 4, 4200, [(100, 168), (50, 84), (25, 42)], [(50, 84)]),
        'cfg5_ext': (1, 16, 32, 4, 22050, [(50, 84)], [(100, 168), (50, 84), (25, 42)]),
        # Mask2Former pixel decoder (seg/configs/_base_/models/mask2former_beit.py:41-51: 8 heads x 32, 3 levels,
        # Lq = S): every token of the three maps queries all three maps; batch 2 = per-batch reference points
        'cfg5_pixdec': (2, 8, 32, 4, 22050, [(100, 168), (50, 84), (25, 42)], [(100, 168), (50, 84), (25, 42)]),
    }
    return table[cfg]
