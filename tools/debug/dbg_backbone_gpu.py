import sys, json, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'vit-adapter_amd'))
import numpy as np, torch
from oracle import backbone_cases as bc, seeded, msda
import ops.modules.ms_deform_attn as mod
from vitadapter.backbones import ViTAdapter
torch.backends.cuda.matmul.allow_tf32=False; torch.backends.cudnn.allow_tf32=False
gold=np.load(os.path.join(ROOT,'tests/golden/backbone.npz'))
HipF = mod.MSDeformAttnFunction
class TorchF:
    @staticmethod
    def apply(value, shapes, lsi, loc, attn, step):
        return msda.core_torch(value, shapes.cpu(), loc, attn)
name=sys.argv[1] if len(sys.argv)>1 else 'det_win_96x128'
case=bc.FULL_CASES[name]
res={}
for tag,F_ in (('hip',HipF),('torch',TorchF)):
    mod.MSDeformAttnFunction=F_
    model=ViTAdapter(**case['cfg'])
    shapes={k:tuple(v.shape) for k,v in model.state_dict().items()}
    model.load_state_dict(seeded.seeded_state_dict(shapes,5))
    model=model.cuda().eval()
    x=bc.full_input(name).cuda().requires_grad_(True)
    outs=model(x)
    gouts=[g.cuda() for g in bc.full_gouts(name,[o.shape for o in outs])]
    sum((o*g).sum() for o,g in zip(outs,gouts)).backward()
    res[tag]=(outs,x.grad,{k:p.grad for k,p in model.named_parameters()})
    w=torch.tensor(gold[name+'_eval_gx']).cuda()
    print(tag,'gx err vs gold',float((x.grad-w).abs().max()), 'ratio', float((x.grad/w).median()))
    for k in range(4):
        print('  f%d err'%(k+1), float((outs[k]-torch.tensor(gold['%s_eval_f%d'%(name,k+1)]).cuda()).abs().max()))
h,t=res['hip'],res['torch']
bad=[(k,float((h[2][k]-t[2][k]).abs().max()),float(t[2][k].abs().max())) for k in h[2] if h[2][k] is not None and float((h[2][k]-t[2][k]).abs().max())>1e-3*max(1,float(t[2][k].abs().max()))]
print('params differing hip vs torch:',len(bad)); print(bad[:20])
