import sys
ROOT='/root/repo'
for p in (ROOT, ROOT+'/mode-2022_amd'): sys.path.insert(0,p)
import numpy as np, torch, torch.nn.functional as F
from mode_hip import functional as HF
dev='cuda:0'
def rnd(shape, seed, scale=1.0): return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape)*scale).astype(np.float32))
for (B,Ci,Co,H,W,dil) in [(4,64,64,16,8,2),(4,64,64,16,8,1),(4,32,32,32,16,1),(2,64,64,16,8,2),(4,64,64,17,9,2),(1,64,64,16,8,2),(4,64,64,32,16,2),(4,128,128,16,8,1)]:
  x=rnd((B,Ci,H,W),1); w=rnd((Co,Ci,3,3),2,0.1); gy=rnd((B,Co,H,W),3)
  xa=x.double().requires_grad_(True); wa=w.double().requires_grad_(True)
  y=F.conv2d(xa,wa,None,1,dil,dil); y.backward(gy.double())
  xd,wd,gd=x.to(dev),w.to(dev),gy.to(dev)
  out=[]
  for a in ('f32','bf16x6'):
    HF.set_conv_arith(a)
    e1=float((HF.conv2d_fwd(xd,wd,dil).cpu().double()-y.detach()).abs().max())
    e2=float((HF.conv2d_bwd_data(gd,wd,dil).cpu().double()-xa.grad).abs().max())
    e3=float((HF.conv2d_bwd_weight(gd,xd,dil).cpu().double()-wa.grad).abs().max())
    out.append('%s fwd %.2e bwd_data %.2e bwd_weight %.2e'%(a,e1,e2,e3))
  print((B,Ci,Co,H,W,dil),' | '.join(out))
