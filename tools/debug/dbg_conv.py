import torch, torch.nn as nn, torch.nn.functional as F, sys
torch.manual_seed(0)
def test(mod, shape, tag):
    x=torch.randn(*shape)
    m_c=mod
    import copy
    m_g=copy.deepcopy(mod).cuda()
    xc=x.clone().requires_grad_(True); xg=x.clone().cuda().requires_grad_(True)
    yc=m_c(xc); yg=m_g(xg)
    r=torch.randn_like(yc)
    (yc*r).sum().backward(); (yg*r.cuda()).sum().backward()
    print('%-28s fwd %.2e  gx %.2e (max %.2e)'%(tag, float((yc-yg.cpu()).abs().max()), float((xc.grad-xg.grad.cpu()).abs().max()), float(xc.grad.abs().max())), end='')
    for (n,pc),(_,pg) in zip(m_c.named_parameters(), m_g.named_parameters()):
        print('  g[%s] %.2e'%(n, float((pc.grad-pg.grad.cpu()).abs().max())), end='')
    print()
for B,H,W in ((1,96,128),(2,96,128),(1,64,64),(2,64,64),(1,128,128),(2,1024,1024)):
    print('--- input',B,H,W)
    test(nn.Conv2d(3,64,16,16), (B,3,H,W), 'patch_embed conv16s16')
    test(nn.Conv2d(3,16,3,2,1,bias=False), (B,3,H,W), 'stem conv3x3 s2 3->16')
    test(nn.Conv2d(3,64,3,2,1,bias=False), (B,3,H,W), 'stem conv3x3 s2 3->64')
    test(nn.Conv2d(16,16,3,1,1,bias=False), (B,16,H//2,W//2), 'conv3x3 s1 16->16')
    test(nn.Conv2d(64,64,3,1,1,bias=False), (B,64,H//2,W//2), 'conv3x3 s1 64->64')
    test(nn.MaxPool2d(3,2,1), (B,16,H//2,W//2), 'maxpool')
    test(nn.Conv2d(64,128,3,2,1,bias=False), (B,64,H//4,W//4), 'conv3x3 s2 64->128')
    if H<1024:
        test(nn.BatchNorm2d(16).eval(), (B,16,H//2,W//2), 'bn eval')
        test(nn.BatchNorm2d(16).train(), (B,16,H//2,W//2), 'bn train')
