import sys, os
ROOT='/root/repo'
for p in (ROOT, ROOT+'/mode-2022_amd', ROOT+'/tests', ROOT+'/tests/golden'): sys.path.insert(0,p)
import numpy as np, torch
import recipe
from oracle import mode_ref
import models, mode_hip
from mode_hip import functional as HF
import test_gpu_parity as T
def run(tag, arith, conv2d_split=True):
  z=np.load(ROOT+'/tests/golden/model_wc_%s.npz'%tag, allow_pickle=False)
  HF.set_conv_arith(arith)
  if not conv2d_split:
    HF._saved = mode_hip.lib().mode_conv2d_split_supported
  net,left,right,gt,seed=T._load(z)
  net.train()
  preds=net(left,right)
  loss=mode_ref.training_loss(preds,gt,~torch.isnan(gt)); loss.backward()
  grads=dict(net.named_parameters()); own=z['truth64/grad_rel_l2']; K=z['train/grad_proj'].shape[1]
  rows=[]
  for i,name in enumerate(z['train/grad_names']):
    name=str(name); p=grads[name]; g=p.grad.detach().cpu().reshape(-1).double().numpy(); norm=float(z['train/grad_norm'][i])
    proj=recipe.projection_signs(seed,i,g.size,K).astype(np.float64)@g
    rel=float(np.sqrt(np.mean((proj-z['train/grad_proj'][i])**2)))/(norm+1e-300)
    bound=max(T._grad_floor(name,p.dim(),tag=='tiny'),5*float(own[i]))
    rows.append((rel/bound,rel,float(own[i]),name))
  rows.sort(reverse=True)
  print(tag,arith,'worst ratio to bound (fail above 1.5):')
  for r in rows[:6]: print('   %.2f  rel %.2e own %.2e %s'%r)
for tag in ('tiny','cfg1'):
  for a in ('f32','bf16x6'):
    run(tag,a)
