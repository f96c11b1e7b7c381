#!/usr/bin/env python
"""Experiment: the bench step (base_det, 1024^2, batch 2, fwd+bwd+AdamW, bf16 autocast) eager vs replayed as one HIP graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from vitadapter.backbones.vit_adapter import PRESETS, ViTAdapter  # noqa: E402


def main():
    args = bench.parse()
    bench.setup_gemm_tuning(args)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = ViTAdapter(**dict(PRESETS['base_det'])).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.05, fused=True, capturable=True)
    x = torch.randn(2, 3, 1024, 1024, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            feats = model(x)
        loss = sum(f.float().mean() for f in feats)
        loss.backward()
        opt.step()
        return loss

    def timed(fn, n=10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(4):
            step()
    torch.cuda.current_stream().wait_stream(s)
    print('eager  %.2f ms/step (host %.2f)' % timed(step), flush=True)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = step()
    print('graph  %.2f ms/step (host %.2f)  loss %.5f' % (*timed(g.replay), float(loss)), flush=True)


if __name__ == '__main__':
    main()
