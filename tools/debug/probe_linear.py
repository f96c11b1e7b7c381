import time, torch
d='cuda'
a=torch.randn(8192,768,device=d,dtype=torch.bfloat16); g=torch.randn(8192,3072,device=d,dtype=torch.bfloat16)
try:
    o=torch.mm(g.t(),a,out_dtype=torch.float32); print('mm out_dtype ok',o.dtype,o.shape)
    ref=(g.t().float()@a.float()); print('err',(o-ref).abs().max().item(), ref.abs().max().item())
    def t(f,n=50):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
    print('bf16 out us',t(lambda: torch.mm(g.t(),a)),'fp32 out us',t(lambda: torch.mm(g.t(),a,out_dtype=torch.float32)))
except Exception as e: print('mm out_dtype FAIL',repr(e))
ps=[torch.randn(768,768*(1+i%4),device=d) for i in range(200)]
bs=[torch.empty_like(p,dtype=torch.bfloat16) for p in ps]
def t(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print('foreach_copy us',t(lambda: torch._foreach_copy_(bs,ps)))
print('loop copy us',t(lambda: [b.copy_(p) for b,p in zip(bs,ps)]))
print('ok',all(torch.equal(b,p.bfloat16()) for b,p in zip(bs,ps)))
x=torch.randn(8192,3072,device=d,dtype=torch.bfloat16)
print('colsum torch us',t(lambda: x.sum(0)), 'float', t(lambda: x.sum(0,dtype=torch.float32)))
