#!/usr/bin/env python
"""Where the HOST time of one bench step goes (cProfile over 5 eager steps of base_det 1024^2, fwd+bwd+AdamW)."""
import cProfile
import os
import pstats
import sys

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
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.05, fused=True)
    x = torch.randn(2, 3, 1024, 1024, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            feats = model(x)
        loss = sum(f.float().mean() for f in feats)
        loss.backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(35)
    st.sort_stats('cumulative').print_stats(45)


if __name__ == '__main__':
    main()
