import sys, os, copy
import torch, torch.nn as nn
torch.manual_seed(0)
for B,H,W in ((1,96,128),(2,96,128),(1,64,64)):
    conv_a=nn.Conv2d(3,16,3,2,1,bias=False); conv_b=nn.Conv2d(3,64,16,16)
    ag,bg=copy.deepcopy(conv_a).cuda(),copy.deepcopy(conv_b).cuda()
    x=torch.randn(B,3,H,W)
    xc=x.clone().requires_grad_(True); xg=x.clone().cuda().requires_grad_(True)
    ya,yb=conv_a(xc),conv_b(xc); ra,rb=torch.randn_like(ya),torch.randn_like(yb)
    ((ya*ra).sum()+(yb*rb).sum()).backward()
    ya2,yb2=ag(xg),bg(xg)
    ga,gb=torch.autograd.grad((ya2*ra.cuda()).sum()+(yb2*rb.cuda()).sum(),[ya2,yb2],retain_graph=True)
    g1=torch.autograd.grad(ya2,[xg],ga,retain_graph=True)[0]
    g2=torch.autograd.grad(yb2,[xg],gb,retain_graph=True)[0]
    print((B,H,W),'branch grads strides',g1.stride(),g2.stride(), 'contig', g1.is_contiguous(), g2.is_contiguous())
    ((ya2*ra.cuda()).sum()+(yb2*rb.cuda()).sum()).backward()
    print('   joint err %.3e   manual sum err %.3e'%(float((xc.grad-xg.grad.cpu()).abs().max()), float((xc.grad-(g1+g2).cpu()).abs().max())))
