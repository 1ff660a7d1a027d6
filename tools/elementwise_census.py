#!/usr/bin/env python3
"""Which aten operators (copies, fills, adds ...) does one eager training step of the benchmark launch beside the library's own
kernels?  torch.profiler with shapes, grouped by (operator, input shapes), sorted by device time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import models  # noqa: E402
import mode_hip  # noqa: E402
from mode_hip import data_parallel  # noqa: E402

mode_hip.lib()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = models.ModeDisparity(192, 'Sphere', 1024, 512, 'Cassini').to(dev).train()
reducer = data_parallel.GradAllReducer(net)
left = torch.randn(2, 3, 1024, 512, device=dev)
right = torch.roll(left, -5, 3)
gt = torch.rand(2, 1024, 512, device=dev) * 100
count = data_parallel.global_valid_count(~torch.isnan(gt))


def step():
  reducer.zero_grad()
  mask = ~torch.isnan(gt)
  gt0 = torch.nan_to_num(gt)
  o = net(left, right)
  loss = 0
  for wgt, p in zip((0.5, 0.7, 1.0), o):
    loss = loss + wgt * data_parallel.global_masked_mean(F.smooth_l1_loss(p, gt0, reduction='none'), mask, count=count)
  loss.backward()


step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
  step()
  torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4):
  dt = getattr(e, 'self_device_time_total', None)
  if dt is None:
    dt = getattr(e, 'self_cuda_time_total', 0)
  if dt > 0 and e.key.startswith('aten::'):
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:90], [s for s in e.stack if 'mode-2022_amd' in s or 'tools/' in s][:2]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('aten operators with device time in one eager step: %.2f ms in %d launches' % (tot / 1e3, sum(r[1] for r in rows)))
for dt, n, k, sh, st in rows[:70]:
  print('%8.1f us %4d  %-22s %-90s %s' % (dt, n, k, sh, ' <- '.join(s.split('mode-2022_amd/')[-1][:60] for s in st)))
