#!/usr/bin/env python3
"""Where do the small launches of a training step come from?  One profiled eager step (torch.profiler, Python stacks),
grouped by (aten op, innermost Python frames of this repo).  Usage: python tools/launch_audit.py [--batch 2]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import models  # noqa: E402
from mode_hip import data_parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--height', type=int, default=1024)
ap.add_argument('--width', type=int, default=512)
ap.add_argument('--maxdisp', type=int, default=192)
args = ap.parse_args()
dev = torch.device('cuda', 0)
net = models.ModeDisparity(args.maxdisp, 'Sphere', args.height, args.width, 'Cassini').to(dev).train()
red = data_parallel.GradAllReducer(net)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
left = torch.randn(args.batch, 3, args.height, args.width, device=dev)
right = torch.randn_like(left)
gt = torch.rand(args.batch, 1, args.height, args.width, device=dev) * 90
mask = gt > 4
count = data_parallel.global_valid_count(mask)


def step():
  red.zero_grad()
  outs = net(left, right)
  loss = sum(w * data_parallel.global_masked_mean(F.smooth_l1_loss(o, gt, reduction='none'), mask, count=count)
             for w, o in zip((0.5, 0.7, 1.0), outs))
  loss.backward()
  opt.step()


for _ in range(2):
  step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
  step()
  torch.cuda.synchronize()

# kernel launches per (op, shapes, repo frame)
ev = prof.events()
launch = collections.Counter()
dur = collections.Counter()
for e in ev:
  if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
    frames = [s for s in (e.stack or []) if 'mode-2022_amd' in s or 'bench' in s or 'launch_audit' in s]
    where = frames[0].split('mode-2022_amd/')[-1] if frames else '(autograd engine / no python frame)'
    shapes = str(e.input_shapes)[:70]
    key = (e.name, shapes, where[:90])
    launch[key] += len(e.kernels)
    dur[key] += sum(k.duration for k in e.kernels)
total = sum(launch.values())
print('kernel launches in one step: %d' % total)
print('%6s %9s  %s' % ('count', 'gpu us', 'op | shapes | where'))
for key, n in launch.most_common(70):
  print('%6d %9.0f  %s | %s | %s' % (n, dur[key], key[0], key[1], key[2]))
by_op = collections.Counter()
for (name, _, _), n in launch.items():
  by_op[name] += n
print('\nby op:')
for name, n in by_op.most_common(40):
  print('%6d  %s' % (n, name))
