import sys, json, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'vit-adapter_amd'))
import numpy as np, torch, torch.nn.functional as F
from oracle import backbone_cases as bc, seeded, msda
import ops.modules.ms_deform_attn as mod
from vitadapter.backbones import ViTAdapter
from vitadapter.backbones.adapter_modules import deform_inputs
torch.backends.cuda.matmul.allow_tf32=False; torch.backends.cudnn.allow_tf32=False
class TorchF:
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, step):
        return msda.core_torch(value, shapes.cpu(), loc, attn)
mod.MSDeformAttnFunction=TorchF
name=sys.argv[1] if len(sys.argv)>1 else 'det_win_96x128'
case=bc.FULL_CASES[name]
def run(dev):
    model=ViTAdapter(**case['cfg'])
    shapes={k:tuple(v.shape) for k,v in model.state_dict().items()}
    model.load_state_dict(seeded.seeded_state_dict(shapes,5))
    model=model.to(dev).eval()
    x0=bc.full_input(name).to(dev).requires_grad_(True)
    T={}
    def keep(n,t):
        t.retain_grad(); T[n]=t; return t
    self=model
    d1,d2=deform_inputs(x0)
    c1,c2,c3,c4=self.spm(x0)
    keep('c1',c1); keep('c2',c2); keep('c3',c3); keep('c4',c4)
    c2,c3,c4=self._add_level_embed(c2,c3,c4)
    n2,n3=c2.size(1),c3.size(1)
    c=torch.cat([c2,c3,c4],1)
    x,H,W=self.patch_embed(x0)
    keep('pe',x)
    bs,n,dim=x.shape
    x=self.pos_drop(x+self._get_pos_embed(self.pos_embed[:,1:],H,W))
    for i,layer in enumerate(self.interactions):
        lo,hi=self.interaction_indexes[i][0],self.interaction_indexes[i][-1]
        x,c=layer(x,c,self.blocks[lo:hi+1],d1,d2,H,W)
        keep('x%d'%i,x); keep('cc%d'%i,c)
    c2=c[:,:n2].transpose(1,2).reshape(bs,dim,H*2,W*2).contiguous()
    c3=c[:,n2:n2+n3].transpose(1,2).reshape(bs,dim,H,W).contiguous()
    c4=c[:,n2+n3:].transpose(1,2).reshape(bs,dim,H//2,W//2).contiguous()
    up=keep('up',self.up(c2))
    c1=up+c1
    xm=keep('xm',x.transpose(1,2).reshape(bs,dim,H,W).contiguous())
    i1=keep('i1',F.interpolate(xm,scale_factor=4,mode='bilinear',align_corners=False))
    i2=keep('i2',F.interpolate(xm,scale_factor=2,mode='bilinear',align_corners=False))
    i4=keep('i4',F.interpolate(xm,scale_factor=0.5,mode='bilinear',align_corners=False))
    c1,c2,c3,c4=c1+i1,c2+i2,c3+xm,c4+i4
    keep('s1',c1);keep('s2',c2);keep('s3',c3);keep('s4',c4)
    outs=[self.norm1(c1),self.norm2(c2),self.norm3(c3),self.norm4(c4)]
    gouts=[g.to(dev) for g in bc.full_gouts(name,[o.shape for o in outs])]
    sum((o*g).sum() for o,g in zip(outs,gouts)).backward()
    T['x0']=x0
    return {k:(v.detach().cpu(), v.grad.detach().cpu()) for k,v in T.items()}
a=run('cpu'); b=run('cuda')
for k in a:
    print('%-5s val err %.3e  grad err %.3e (max %.3e)'%(k,float((a[k][0]-b[k][0]).abs().max()),float((a[k][1]-b[k][1]).abs().max()),float(a[k][1].abs().max())))
print('---- split x0 grad by branch')
def run2(dev, grads_from):
    model=ViTAdapter(**case['cfg'])
    shapes={k:tuple(v.shape) for k,v in model.state_dict().items()}
    model.load_state_dict(seeded.seeded_state_dict(shapes,5))
    model=model.to(dev).eval()
    x0=bc.full_input(name).to(dev).requires_grad_(True)
    cs=model.spm(x0)
    pe,H,W=model.patch_embed(x0)
    g_spm=torch.autograd.grad(cs,[x0],[grads_from[k][1].to(dev) for k in ('c1','c2','c3','c4')],retain_graph=True)[0]
    g_pe=torch.autograd.grad([pe],[x0],[grads_from['pe'][1].to(dev)],retain_graph=True)[0]
    return g_spm.cpu(), g_pe.cpu()
sc,pc=run2('cpu',a); sg,pg=run2('cuda',a)
print('spm branch err %.3e (max %.3e)  pe branch err %.3e (max %.3e)'%(float((sc-sg).abs().max()),float(sc.abs().max()),float((pc-pg).abs().max()),float(pc.abs().max())))
print('sum vs cpu x0.grad: cpu-sum %.3e  gpu-sum %.3e ; gpu x0.grad vs gpu-sum %.3e'%(float((sc+pc-a['x0'][1]).abs().max()), float((sg+pg-a['x0'][1]).abs().max()), float((sg+pg-b['x0'][1]).abs().max())))
