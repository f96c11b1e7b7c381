import sys, os, copy
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'vit-adapter_amd'))
import torch, torch.nn as nn
from oracle import seeded, backbone_cases as bc
from vitadapter.backbones.adapter_modules import SpatialPriorModule
name='det_win_96x128'
def run(dev, inplace=True):
    spm=SpatialPriorModule(16,64)
    sd={k:tuple(v.shape) for k,v in spm.state_dict().items()}
    spm.load_state_dict(seeded.seeded_state_dict({('spm.'+k):s for k,s in sd.items()},5) and {k:seeded.seeded_param('spm.'+k,s,5) for k,s in sd.items()})
    spm=spm.to(dev).eval()
    if not inplace:
        for m in spm.modules():
            if isinstance(m, nn.ReLU): m.inplace=False
    x=bc.full_input(name).to(dev).requires_grad_(True)
    acts={}
    h=x
    for i,l in enumerate(spm.stem):
        h=l(h)
        if not (isinstance(l, nn.ReLU) and inplace):
            h.retain_grad(); acts['stem%d'%i]=h
    c1=h
    c2=spm.conv2(c1); c3=spm.conv3(c2); c4=spm.conv4(c3)
    for n,t in (('c2',c2),('c3',c3),('c4',c4)): t.retain_grad(); acts[n]=t
    outs=[spm.fc1(c1),spm.fc2(c2),spm.fc3(c3),spm.fc4(c4)]
    gs=[seeded.randn('dbg/g%d'%k,o.shape,1).to(dev) for k,o in enumerate(outs)]
    sum((o*g).sum() for o,g in zip(outs,gs)).backward()
    acts['x']=x
    return {k:(v.detach().cpu(),v.grad.detach().cpu()) for k,v in acts.items()}
for inplace in (True,False):
    a=run('cpu',inplace); b=run('cuda',inplace)
    print('inplace',inplace)
    for k in a:
        print('  %-7s val err %.2e grad err %.2e (max %.2e)'%(k,float((a[k][0]-b[k][0]).abs().max()),float((a[k][1]-b[k][1]).abs().max()),float(a[k][1].abs().max())))
print('--- maxpool isolated on the real activations')
import torch.nn.functional as F
a=run('cpu',False)
inp=a['stem8'][0]; gout=a['stem9'][1]
ic=inp.clone().requires_grad_(True); ig=inp.clone().cuda().requires_grad_(True)
oc=F.max_pool2d(ic,3,2,1); og=F.max_pool2d(ig,3,2,1)
oc.backward(gout); og.backward(gout.cuda())
d=(ic.grad-ig.grad.cpu()).abs()
print('maxpool bwd err', float(d.max()), 'n bad', int((d>1e-4).sum()))
idx=torch.nonzero(d>1e-4)[:4]
for i in idx:
    n,c,y,x=i.tolist()
    print('pos',(n,c,y,x),'val',float(inp[n,c,y,x]),'cpu g',float(ic.grad[n,c,y,x]),'gpu g',float(ig.grad[n,c,y,x]))
    y0,x0=max(y-2,0),max(x-2,0)
    print(inp[n,c,y0:y+3,x0:x+3])
