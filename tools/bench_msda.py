#!/usr/bin/env python
"""Kernel micro-benchmark: MSDA fwd / bwd at every BASELINE call shape, HIP-event timed.
Prints one line per (config, mode, direction): time, algorithmic GB/s, fraction of 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vit-adapter_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import MultiScaleDeformableAttention as MSDA  # noqa: E402
from test_msda_gpu import _full_inputs  # noqa: E402
from oracle import cases  # noqa: E402


def timeit(fn, iters=20, warm=3):
    """Median over ``iters`` of the HIP-event time of one call (seconds).  The median, not total / iters: one stall
    inside the loop (an allocator refill, a clock ramp: 30-60 ms, seen about once per run of this script) would
    otherwise land in whichever row it hit (2947 us instead of 190 us)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    times = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return times[len(times) // 2] * 1e-3


def main():
    cfgs = sys.argv[1:] or ['cfg1', 'cfg2_inj', 'cfg2_ext', 'cfg3_inj', 'cfg3_ext', 'cfg4_inj',
                            'cfg4_ext', 'cfg5_inj', 'cfg5_ext', 'cfg5_pixdec']
    for cfg in cfgs:
        N, M, D, P, Lq, shapes, _ = cases.bench_inputs(cfg)
        L, S = len(shapes), sum(h * w for h, w in shapes)
        fb = 4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D)
        bb = 4 * (2 * N * S * M * D + 6 * N * Lq * M * L * P + N * Lq * M * D)
        for mode in (os.environ.get('MODES', 'uniform,adapter').split(',')):
            v, s, i, l, a, g = _full_inputs(cfg, mode)
            tf = timeit(lambda: MSDA.ms_deform_attn_forward(v, s, i, l, a, 64))
            tb = timeit(lambda: MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64))
            # backward: the tile pass stores grad_value (no zero-fill)
            print('%-9s %-8s fwd %8.1f us %7.1f GB/s (%.3f of 8TB/s) | bwd %8.1f us %7.1f GB/s (%.3f)'
                  % (cfg, mode, tf * 1e6, fb / tf / 1e9, fb / tf / 8e12, tb * 1e6, bb / tb / 1e9,
                     bb / tb / 8e12), flush=True)


if __name__ == '__main__':
    main()
