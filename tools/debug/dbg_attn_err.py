"""Error of the bf16 attention forward against fp32 math on the same bf16 inputs, small and large N."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
    sys.path.insert(0, p)
import torch
from vitadapter import kernels

for B, N, H in [(2, 16, 6), (2, 196, 3), (1, 1024, 4), (2, 4096, 2)]:
    torch.manual_seed(0)
    qkv = (torch.randn(B, N, 3, H, 64, device='cuda') * 1.5).to(torch.bfloat16)
    out = kernels.attention(qkv, 0.125).float()
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4).unbind(0)
    ref = (((q @ k.transpose(-2, -1)) * 0.125).softmax(-1) @ v).transpose(1, 2)
    kernels.FLAGS['force_math_attention'] = True
    mth = kernels.attention(qkv, 0.125).float()
    kernels.FLAGS['force_math_attention'] = False
    print(B, N, H, 'kernel: max %.4f relL2 %.5f | bf16 math: max %.4f relL2 %.5f' % (
        (out - ref).abs().max(), (out - ref).norm() / ref.norm(), (mth - ref).abs().max(), (mth - ref).norm() / ref.norm()))
