import sys
ROOT='/root/repo'
for p in (ROOT, ROOT+'/mode-2022_amd', ROOT+'/tests', ROOT+'/tests/golden'): sys.path.insert(0,p)
import numpy as np, torch
import models, mode_hip
from mode_hip import functional as HF
import test_gpu_parity as T
z=np.load(ROOT+'/tests/golden/model_wc_tiny.npz', allow_pickle=False)
outs={}
for a in ('f32','bf16x6'):
  HF.set_conv_arith(a)
  net,left,right,gt,seed=T._load(z); net.train()
  blk=net.feature_extraction.layer3[2]
  print(type(blk).__name__, [n for n,_ in blk.named_children()])
  conv=blk.conv1[0][0] if isinstance(blk.conv1[0], torch.nn.Sequential) else blk.conv1[0]
  bn=blk.conv1[0][1] if isinstance(blk.conv1[0], torch.nn.Sequential) else blk.conv1[1]
  store={}
  from models import stage3d
  orig=stage3d.bn_act
  def spy(bnm, x, *args, **kw):
    if bnm is bn:
      store['y']=x.detach().cpu().double()
    return orig(bnm, x, *args, **kw)
  stage3d.bn_act=spy
  try:
    net(left,right)
  finally:
    stage3d.bn_act=orig
  if 'y' not in store:
    print('spy did not fire'); continue
  y=store['y']
  # the extractor runs on [left; right] with per-group statistics: normalise per half
  pre=[]
  for grp in y.chunk(2,0):
    m=grp.mean((0,2,3),keepdim=True); v=grp.var((0,2,3),unbiased=False,keepdim=True)
    pre.append((grp-m)/torch.sqrt(v+bn.eps)*bn.weight.detach().cpu().double().view(1,-1,1,1)+bn.bias.detach().cpu().double().view(1,-1,1,1))
  outs[a]=torch.cat(pre,0)
if len(outs)==2:
  p0,p1=outs['f32'],outs['bf16x6']
  print('elements', p0.numel(), 'max |diff| of the pre-activation', float((p0-p1).abs().max()))
  flips=(p0>0)!=(p1>0)
  print('ReLU mask disagreements:', int(flips.sum()), 'values there:', p0[flips][:8].tolist(), p1[flips][:8].tolist())
  print('smallest |pre-activation|:', float(p0.abs().min()))
