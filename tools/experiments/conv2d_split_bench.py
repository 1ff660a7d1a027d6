import sys, os, time
ROOT='/root/repo'
for p in (ROOT, ROOT+'/mode-2022_amd'): sys.path.insert(0,p)
import torch
from mode_hip import functional as HF
dev='cuda:0'
for (B,Ci,Co,H,W,dil) in [(4,64,64,256,128,1),(4,64,64,512,256,1),(4,64,64,256,128,2),(4,128,128,256,128,1),(4,32,32,512,256,1)]:
  x=torch.randn(B,Ci,H,W,device=dev); w=torch.randn(Co,Ci,3,3,device=dev)*0.05
  for a in ('f32','bf16x6'):
    HF.set_conv_arith(a)
    for fn,name in ((lambda: HF.conv2d_fwd(x,w,dil),'fwd'),(lambda: HF.conv2d_bwd_data(x,w,dil),'bwd_data'),(lambda: HF.conv2d_bwd_weight(x,x,dil),'bwd_weight')):
      if Ci!=Co and name=='bwd_data': continue
      for _ in range(3): fn()
      torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20): fn()
      e1.record(); torch.cuda.synchronize()
      ms=e0.elapsed_time(e1)/20
      print('conv2d %s %d->%d d%d %dx%d B=%d %-7s %.3f ms %.1f TF'%(name,Ci,Co,dil,H,W,B,a,ms,2*9*Ci*Co*B*H*W/ms/1e9))
