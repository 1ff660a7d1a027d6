#!/usr/bin/env python3
"""Development probe (CPU): how well conditioned is ModeDisparity at a given state?  E32 = max|fp32 - fp64| of the oracle on the
same state/inputs, for the recipe state and for states after a few Adam steps on a constant-shift pair.

  python tools/conditioning_probe.py --size tiny|cfg1 --steps N [--subset bn]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
  sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import recipe  # noqa: E402
from oracle import mode_ref  # noqa: E402


def e32(P, left, right, maxdisp, pos, train):
  P32 = {k: v.detach().clone() for k, v in P.items()}
  P64 = {k: (v.detach().double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
  with torch.no_grad():
    a = mode_ref.mode_disparity(P32, left, right, maxdisp, pos, train)
    b = mode_ref.mode_disparity(P64, left.double(), right.double(), maxdisp, pos, train)
  if not train:
    a, b = (a,), (b,)
  mx = max(float((x.double() - y).abs().max()) for x, y in zip(a, b))
  mn = max(float((x.double() - y).abs().mean()) for x, y in zip(a, b))
  return mx, mn, [float(y.mean()) for y in b], [float(y.std()) for y in b]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--size', default='tiny')
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--every', type=int, default=5)
  ap.add_argument('--subset', default='all')
  ap.add_argument('--lr', type=float, default=1e-3)
  ap.add_argument('--shift', type=int, default=3)
  ap.add_argument('--seed', type=int, default=300)
  args = ap.parse_args()
  torch.set_num_threads(8)
  maxdisp, H, W, B = dict(tiny=(16, 64, 32, 2), small=(32, 128, 64, 2), cfg1=(64, 512, 256, 1))[args.size]
  P = recipe.recipe_state(recipe.load_manifest(), args.seed)
  left, right = recipe.recipe_images(B, H, W, args.seed + 1, shift=args.shift)
  gt = torch.full((B, 1, H, W), float(args.shift))
  mask = torch.ones_like(gt, dtype=torch.bool)
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')

  def report(tag):
    t = time.time()
    tr = e32(P, left, right, maxdisp, pos, True)
    print('%s train: E32 max %.3e mean %.3e  pred mean %s std %s  (%.1fs)' % (tag, tr[0], tr[1], np.round(tr[2], 3), np.round(tr[3], 3),
                                                                              time.time() - t), flush=True)

  report('step 0')
  names = [k for k, v in P.items() if v.is_floating_point() and 'running' not in k]
  if args.subset == 'bn':
    names = [k for k in names if P[k].dim() == 1 or k.startswith('classif') and k.endswith('.2.weight')]
  elif args.subset == '3d':
    names = [k for k in names if not k.startswith('feature_extraction')]
  params = []
  for k in names:
    P[k].requires_grad_(True)
    params.append(P[k])
  print('training %d tensors, %d values' % (len(params), sum(p.numel() for p in params)))
  opt = torch.optim.Adam(params, lr=args.lr)
  for it in range(1, args.steps + 1):
    opt.zero_grad()
    preds = mode_ref.mode_disparity(P, left, right, maxdisp, pos, True)
    loss = mode_ref.training_loss(preds, gt, mask)
    loss.backward()
    opt.step()
    print('it %d loss %.4f' % (it, float(loss)), flush=True)
    if it % args.every == 0:
      report('step %d' % it)


if __name__ == '__main__':
  main()
