import sys, os, copy
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'vit-adapter_amd'))
import torch, torch.nn as nn
from vitadapter.backbones.adapter_modules import SpatialPriorModule
torch.manual_seed(0)
def cmp(mod, x, tag, pick=None):
    mg=copy.deepcopy(mod).cuda()
    xc=x.clone().requires_grad_(True); xg=x.clone().cuda().requires_grad_(True)
    yc=mod(xc); yg=mg(xg)
    if pick is not None: yc=yc[pick]; yg=yg[pick]
    if isinstance(yc,(tuple,list)):
        rs=[torch.randn_like(t) for t in yc]
        sum((a*r).sum() for a,r in zip(yc,rs)).backward(); sum((a*r.cuda()).sum() for a,r in zip(yg,rs)).backward()
    else:
        r=torch.randn_like(yc); (yc*r).sum().backward(); (yg*r.cuda()).sum().backward()
    print('%-40s gx err %.2e (max %.2e)'%(tag, float((xc.grad-xg.grad.cpu()).abs().max()), float(xc.grad.abs().max())))
for B,H,W in ((1,96,128),(2,64,64)):
    for mode in ('eval','train'):
        spm=SpatialPriorModule(16,64); spm.train(mode=='train')
        for m in spm.modules():
            if isinstance(m, nn.SyncBatchNorm):
                m.running_mean.normal_(0,0.1); m.running_var.uniform_(0.5,1.5); m.weight.data.normal_(1,0.1); m.bias.data.normal_(0,0.1)
        x=torch.randn(B,3,H,W)
        cmp(spm,x,'spm all %s %s'%(mode,(B,H,W)))
        for k in range(4): cmp(spm,x,'spm out c%d'%(k+1),pick=k)
        cmp(spm.stem,x,'stem only')
        cmp(spm.stem[:3],x,'stem[:3] conv-bn-relu')
        cmp(spm.stem[:2],x,'stem[:2] conv-bn')
        cmp(spm.stem[:6],x,'stem[:6]')
        cmp(spm.stem[:9],x,'stem[:9]')
        bn2d=nn.Sequential(nn.Conv2d(3,16,3,2,1,bias=False), nn.BatchNorm2d(16), nn.ReLU(inplace=True)); bn2d.train(mode=='train')
        cmp(bn2d,x,'conv-BatchNorm2d-relu')
