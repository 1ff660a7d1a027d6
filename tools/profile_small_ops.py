import sys, os, collections
ROOT='/root/repo'
for p in (ROOT, ROOT+'/mode-2022_amd'): sys.path.insert(0,p)
import torch, models
from mode_hip import data_parallel
import bench
dev=torch.device('cuda',0)
net=models.ModeDisparity(192,'Sphere',1024,512,'Cassini').to(dev)
reducer=data_parallel.GradAllReducer(net)
left,right,gt=bench.synthetic_batch(2,1024,512,192,dev,seed=1)
count=data_parallel.global_valid_count(~torch.isnan(gt))
def step():
  reducer.zero_grad()
  preds=net(left,right)
  import torch.nn.functional as F
  mask=~torch.isnan(gt); gt0=torch.where(mask,gt,torch.zeros_like(gt)); loss=0
  for wgt,o in zip((0.5,0.7,1.0),preds):
    loss=loss+wgt*data_parallel.global_masked_mean(F.smooth_l1_loss(o,gt0,reduction='none'),mask,count=count)
  loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
  step()
torch.cuda.synchronize()
c=collections.Counter()
for e in prof.events():
  if e.name in ('aten::zero_','aten::fill_','aten::copy_','aten::zeros','aten::zeros_like','aten::clone','aten::contiguous') :
    shape=str(e.input_shapes)[:60]
    st=[s for s in (e.stack or []) if 'mode-2022_amd' in s or 'bench.py' in s or 'data_parallel' in s]
    c[(e.name, shape, st[0] if st else (e.stack[0] if e.stack else '?'))]+=1
for k,v in c.most_common(40): print(v,k)
