"""Micro-benchmark of the row-streaming kernels (csrc/fused_ops.hip) at the bench shapes, through the
C ABI, HIP-event timed, rotating over enough buffer sets that nothing is served from the 256 MB
infinity cache.  Prints us per call and the rate on the algorithmic bytes.

    python tools/bench_fused.py [rows C]        (default 8192 768 and 43008 768)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vit-adapter_amd'))
import _vah  # noqa: E402

L = _vah.lib


def timeit(fn, sets, iters=40):
    for i in range(len(sets)):
        fn(sets[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    shapes = [(8192, 768), (43008, 768)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
    st = torch.cuda.current_stream().cuda_stream
    d = 'cuda'
    for rows, C in shapes:
        B = 2
        rpb = rows // B
        nsets = max(2, int(600e6 / (rows * C * 18)) + 1)
        sets = []
        for _ in range(nsets):
            s = dict(x=torch.randn(rows, C, device=d), z=torch.randn(rows, C, device=d).bfloat16(),
                     g=torch.randn(rows, C, device=d).bfloat16(), gres=torch.randn(rows, C, device=d),
                     y=torch.empty(rows, C, device=d), h=torch.empty(rows, C, device=d, dtype=torch.bfloat16),
                     dz=torch.empty(rows, C, device=d, dtype=torch.bfloat16),
                     mean=torch.zeros(rows, device=d), rstd=torch.ones(rows, device=d))
            sets.append(s)
        w, b, gamma = torch.ones(C, device=d), torch.zeros(C, device=d), torch.ones(C, device=d)
        sc = torch.ones(B, device=d)
        dw, db, dg = (torch.empty(C, device=d) for _ in range(3))
        ws = torch.empty(L.vah_reduce_ws_floats(2 * C), device=d)
        p = lambda t: t.data_ptr()
        ops = {
            'ln_fwd': (6, lambda s: L.vah_layernorm_fwd_f32_bf16(p(s['x']), p(w), p(b), rows, C, 1e-6, p(s['h']), p(s['mean']), p(s['rstd']), st)),
            'ln_bwd': (10, lambda s: L.vah_layernorm_bwd_f32_bf16(p(s['x']), p(s['g']), p(w), p(s['mean']), p(s['rstd']), None, rows, C, p(s['y']), p(dw), p(db), p(ws), st)),
            'ln_bwd+gres': (14, lambda s: L.vah_layernorm_bwd_f32_bf16(p(s['x']), p(s['g']), p(w), p(s['mean']), p(s['rstd']), p(s['gres']), rows, C, p(s['y']), p(dw), p(db), p(ws), st)),
            'sr_fwd': (10, lambda s: L.vah_scale_residual_fwd(p(s['x']), p(s['z']), p(gamma), p(sc), B, rpb, C, p(s['y']), st)),
            'sr_bwd (no gamma)': (6, lambda s: L.vah_scale_residual_bwd(p(s['gres']), p(s['z']), None, p(sc), B, rpb, C, p(s['dz']), None, None, st)),
            'sr_bwd': (8, lambda s: L.vah_scale_residual_bwd(p(s['gres']), p(s['z']), p(gamma), p(sc), B, rpb, C, p(s['dz']), p(dg), p(ws), st)),
            'res_ln_fwd': (12, lambda s: L.vah_residual_layernorm_fwd(p(s['x']), p(s['z']), p(gamma), p(sc), B, rpb, C, p(w), p(b), 1e-6, p(s['y']), p(s['h']), p(s['mean']), p(s['rstd']), st)),
            'res_ln_bwd': (18, lambda s: L.vah_residual_layernorm_bwd(p(s['x']), p(s['g']), p(w), p(s['mean']), p(s['rstd']), p(s['gres']), p(s['z']), p(gamma), p(sc), B, rpb, C, p(s['y']), p(s['dz']), p(dg), p(dw), p(db), p(ws), st)),
            'colsum': (2, lambda s: L.vah_colsum_bf16(p(s['g']), rows, C, p(dw), p(ws), st)),
            'torch add f32 (ref)': (12, lambda s: torch.add(s['x'], s['gres'], out=s['y'])),
            'torch copy f32->bf16 (ref)': (6, lambda s: s['h'].copy_(s['x'])),
        }
        if rows % 42 == 0:                                  # ConvFFN token tensor: 21 n tokens per image
            Cd, Hd = C // 4, int((rows // B // 21) ** 0.5) * 2
            if 21 * (Hd // 2) ** 2 * B == rows:
                xs = [torch.randn(rows, Cd, device=d).bfloat16() for _ in range(8)]
                ys = [torch.empty(rows, Cd, device=d, dtype=torch.bfloat16) for _ in range(8)]
                wd, bd = torch.randn(Cd, 9, device=d), torch.randn(Cd, device=d)
                dwg = torch.empty(Cd * 10, device=d)
                wsd = torch.empty(L.vah_reduce_ws_floats(10 * Cd), device=d)
                dsets = [dict(x=a, y=b_) for a, b_ in zip(xs, ys)]
                for name, bpe, fn in (
                        ('dwconv fwd (C/4)', 4, lambda s: L.vah_dwconv3x3_tokens_bf16(p(s['x']), p(wd), p(bd), B, Hd, Hd, Cd, 0, p(s['y']), st)),
                        ('dwconv dgrad (C/4)', 4, lambda s: L.vah_dwconv3x3_tokens_bf16(p(s['x']), p(wd), None, B, Hd, Hd, Cd, 1, p(s['y']), st)),
                        ('dwconv wgrad (C/4)', 4, lambda s: L.vah_dwconv3x3_tokens_wgrad_bf16(p(s['x']), p(s['y']), B, Hd, Hd, Cd, p(dwg), p(dwg[Cd * 9:]), p(wsd), st))):
                    us = timeit(fn, dsets)
                    print('  %-28s %7.1f us  %6.2f TB/s' % (name, us, rows * Cd * bpe / us / 1e6))
        print('rows %d C %d, %d buffer sets' % (rows, C, nsets))
        for name, (bpe, fn) in ops.items():
            us = timeit(fn, sets)
            print('  %-28s %7.1f us  %6.2f TB/s' % (name, us, rows * C * bpe / us / 1e6))


if __name__ == '__main__':
    main()
