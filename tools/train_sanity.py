#!/usr/bin/env python3
"""A short real training run on synthetic data at the benchmark size: the loss of the default (all fast paths) configuration must
fall and stay finite over a few dozen Adam steps.  Usage: python tools/train_sanity.py [--steps 30] [--batch 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import models  # noqa: E402
from mode_hip import data_parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--height', type=int, default=1024)
ap.add_argument('--width', type=int, default=512)
ap.add_argument('--maxdisp', type=int, default=192)
a = ap.parse_args()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = models.ModeDisparity(a.maxdisp, 'Sphere', a.height, a.width, 'Cassini').to(dev).train()
red = data_parallel.GradAllReducer(net)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
g = torch.Generator(device='cpu').manual_seed(1)
left = torch.rand(a.batch, 3, a.height, a.width, generator=g).to(dev)
shift = 12
right = torch.roll(left, -shift, 3) + 0.01 * torch.randn(left.shape, generator=g).to(dev)
gt = torch.full((a.batch, 1, a.height, a.width), float(shift), device=dev)
losses = []
for it in range(a.steps):
  red.zero_grad()
  o1, o2, o3 = net(left, right)
  loss = 0.5 * F.smooth_l1_loss(o1, gt) + 0.7 * F.smooth_l1_loss(o2, gt) + F.smooth_l1_loss(o3, gt)
  loss.backward()
  red.all_reduce()
  opt.step()
  losses.append(float(loss))
  if it % 5 == 0 or it == a.steps - 1:
    print('step %3d  loss %.4f' % (it, losses[-1]), flush=True)
assert all(l == l and abs(l) < 1e6 for l in losses), 'non-finite loss'
assert min(losses[-5:]) < 0.5 * losses[0], 'the loss did not fall: %.4f -> %.4f' % (losses[0], losses[-1])
print('ok: loss %.4f -> %.4f over %d steps' % (losses[0], losses[-1], a.steps))
