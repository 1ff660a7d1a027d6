#!/usr/bin/env python3
"""Golden vectors for ModeDisparity(conv='Regular') -- the PSMNet SPP extractor of models/submodule.py:205-268 (SURVEY 8f
rank 4) -- made by the imported reference with the harness stand-ins of make_golden.py.

  python tests/golden/make_golden_regular.py      # writes model_regular.npz

The fp64 "truth" is the reference itself evaluated in float64 (the functional oracle covers the spherical extractor only)."""
import json
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import make_golden as mg
from make_golden import HERE, recipe

CFG = dict(maxdisp=16, H=256, W=256, B=2, seed=300)  # 1/4 resolution 64x64: the largest SPP branch pools 64x64 -> 1x1


def main():
  torch.set_num_threads(8)
  models, _ = mg.import_reference()
  c = CFG
  torch.manual_seed(0)
  m = models.ModeDisparity(c['maxdisp'], 'Regular')
  manifest = mg.load_recipe(m, c['seed'])
  left, right = recipe.recipe_images(c['B'], c['H'], c['W'], c['seed'] + 1)
  gt = recipe.recipe_disparity(c['B'], c['H'], c['W'], c['seed'] + 2, c['maxdisp'])
  mask = ~torch.isnan(gt)
  out = dict(cfg=np.array([c['maxdisp'], c['H'], c['W'], c['B'], c['seed']]), manifest=np.array(json.dumps([[k, list(s)] for k, s in manifest])))
  sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4))
  m.train()
  preds = m(left, right)
  loss = 0.5 * F.smooth_l1_loss(preds[0][mask], gt[mask]) + 0.7 * F.smooth_l1_loss(preds[1][mask], gt[mask]) + F.smooth_l1_loss(preds[2][mask], gt[mask])
  loss.backward()
  out['train/loss'] = np.array(float(loss))
  for i, p in enumerate(preds):
    out['train/pred%d' % (i + 1)] = p.detach()[sub].numpy()
  out.update({'train/' + k: v for k, v in mg.grad_summary(m).items()})
  bns = [x for x in m.modules() if isinstance(x, (nn.BatchNorm2d, nn.BatchNorm3d))]
  for x in bns:
    x.momentum = 1.0
  with torch.no_grad():
    m(left, right)
  for x in bns:
    x.momentum = 0.1
  out.update(mg.bn_stats(m))
  m.eval()
  with torch.no_grad():
    out['eval/pred3'] = m(left, right)[sub].numpy()
  # float64 evaluation by the reference itself
  m64 = models.ModeDisparity(c['maxdisp'], 'Regular').double()
  m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in recipe.recipe_state(manifest, c['seed']).items()})
  # the reference builds its cost volume with torch.FloatTensor (mode_disparity.py:104): for the float64 run only, that
  # name is pointed at the double tensor type (harness-side, like the other stand-ins)
  float_tensor = torch.FloatTensor
  torch.FloatTensor = torch.DoubleTensor
  m64.train()
  with torch.no_grad():
    t = m64(left.double(), right.double())
  for i, p in enumerate(t):
    out['truth64/train_pred%d' % (i + 1)] = p[sub].numpy()
  sd = m64.state_dict()
  for k, v in out.items():
    if k.startswith('bn/'):
      sd[k[3:]] = torch.from_numpy(v).double()
  m64.load_state_dict(sd)
  m64.eval()
  with torch.no_grad():
    out['truth64/eval_pred3'] = m64(left.double(), right.double())[sub].numpy()
  torch.FloatTensor = float_tensor
  np.savez_compressed(os.path.join(HERE, 'model_regular.npz'), **out)
  print('wrote model_regular.npz: loss %.5f; E_ref train %.2e eval %.2e' %
        (float(loss), max(np.abs(out['train/pred%d' % i] - out['truth64/train_pred%d' % i]).max() for i in (1, 2, 3)),
         np.abs(out['eval/pred3'] - out['truth64/eval_pred3']).max()))


if __name__ == '__main__':
  main()
