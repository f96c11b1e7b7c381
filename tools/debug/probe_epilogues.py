import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'vit-adapter_amd'))
import _vah
from vitadapter import fused
import torch.nn.functional as F
d = 'cuda'
R, K, N = 8192, 768, 3072
x = torch.randn(R, K, device=d).bfloat16(); w = (torch.randn(N, K, device=d) * 0.03).bfloat16(); b = torch.randn(N, device=d)
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
# 1. GELU_AUX_BIAS
try:
    aux = torch.empty(R, N, device=d, dtype=torch.bfloat16)
    y = fused.gemm_bf16(x, w, trans_b=True, bias=b, epilogue=_vah.GEMM_EPI_BIAS_GELU_AUX, aux=aux)
    pre = (x.float() @ w.float().t() + b)
    print('gelu_aux: aux err', (aux.float() - pre).abs().max().item(), 'y err vs erf-gelu', (y.float() - F.gelu(pre)).abs().max().item(),
          'vs tanh-gelu', (y.float() - F.gelu(pre, approximate='tanh')).abs().max().item())
    print('  us fused', t(lambda: fused.gemm_bf16(x, w, trans_b=True, bias=b, epilogue=_vah.GEMM_EPI_BIAS_GELU_AUX, aux=aux)),
          'us plain+gelu', t(lambda: F.gelu(fused.gemm_bf16(x, w, trans_b=True, bias=b))))
except Exception as e: print('gelu_aux FAIL', e)
# 2. DGELU: dx_pre = (g @ W2) * gelu'(aux)   g [R, K2] W2 [K2, N]
try:
    g = torch.randn(R, K, device=d).bfloat16(); w2 = (torch.randn(K, N, device=d) * 0.03).bfloat16()
    aux = torch.randn(R, N, device=d).bfloat16()
    o = fused.gemm_bf16(g, w2, epilogue=_vah.GEMM_EPI_DGELU, aux=aux)
    a32 = aux.float().requires_grad_(True); F.gelu(a32).backward(g.float() @ w2.float())
    print('dgelu err', (o.float() - a32.grad).abs().max().item(), a32.grad.abs().max().item())
    print('  us fused', t(lambda: fused.gemm_bf16(g, w2, epilogue=_vah.GEMM_EPI_DGELU, aux=aux)), 'us plain', t(lambda: fused.gemm_bf16(g, w2)))
except Exception as e: print('dgelu FAIL', e)
# 3. BGRAD on wgrad
for f32 in (True, False):
    try:
        g = torch.randn(R, N, device=d).bfloat16()
        bg = torch.zeros(N, device=d, dtype=torch.float32)
        o = fused.gemm_bf16(g, x, trans_a=True, out_dtype=torch.float32 if f32 else torch.bfloat16, bias=bg, epilogue=_vah.GEMM_EPI_BGRAD_A)
        print('bgrad f32=%d: dw err' % f32, (o.float() - g.float().t() @ x.float()).abs().max().item(), 'db err', (bg - g.float().sum(0)).abs().max().item())
        print('  us fused', t(lambda: fused.gemm_bf16(g, x, trans_a=True, out_dtype=torch.float32 if f32 else torch.bfloat16, bias=bg, epilogue=_vah.GEMM_EPI_BGRAD_A)),
              'plain', t(lambda: fused.gemm_bf16(g, x, trans_a=True, out_dtype=torch.float32 if f32 else torch.bfloat16)))
    except Exception as e: print('bgrad FAIL f32=%d' % f32, e)
# 4. transposed-g wgrad: gT [N, R] @ x [R, K]
gT = torch.randn(N, R, device=d).bfloat16()
print('wgrad from gT us', t(lambda: fused.gemm_bf16(gT, x, out_dtype=torch.float32)), ' from g (trans_a) us', t(lambda: fused.gemm_bf16(g, x, trans_a=True, out_dtype=torch.float32)))
print('transpose us', t(lambda: g.t().contiguous()))
