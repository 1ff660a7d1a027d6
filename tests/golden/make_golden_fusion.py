#!/usr/bin/env python3
"""Golden vectors for the fusion stage, made by the REFERENCE's models/mode_fusion.py in the dev container.

  python tests/golden/make_golden_fusion.py     # writes manifest_mode_fusion.json and fusion_tiny.npz

No stand-ins are needed: the reference file imports torch and numpy only.  It is loaded by path (importing the reference's
`models` package would pull in the CUDA extension).  Weights follow tests/golden/recipe.py, inputs are seeded noise; the
fp64 evaluation of the same network is stored next to the reference's fp32 output (as for the disparity stage)."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402

REF_FILE = '/root/reference/models/mode_fusion.py'
TINY = dict(maxdepth=10.0, channels=[8, 16, 32, 64], B=2, H=64, W=32, seed=77)


def fusion_inputs(B, H, W, seed, maxdepth):
  rs = np.random.RandomState(seed)
  depthes = [torch.from_numpy((rs.rand(B, 1, H, W) * maxdepth).astype(np.float32)) for _ in range(6)]
  confs = [torch.from_numpy(rs.rand(B, 1, H, W).astype(np.float32)) for _ in range(6)]
  rgbs = [torch.from_numpy(rs.rand(B, 3, H, W).astype(np.float32)) for _ in range(4)]
  gt = torch.from_numpy((rs.rand(B, H, W) * maxdepth * 1.1).astype(np.float32))  # some above maxdepth: masked out
  return depthes, confs, rgbs, gt


def main():
  spec = importlib.util.spec_from_file_location('ref_mode_fusion', REF_FILE)
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)
  torch.manual_seed(0)
  full = ref.ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12})
  with open(os.path.join(HERE, 'manifest_mode_fusion.json'), 'w') as f:
    json.dump([[k, list(v.shape)] for k, v in full.state_dict().items()], f)

  t = TINY
  net = ref.ModeFusion(t['maxdepth'], t['channels'], {'depth': 12, 'rgb': 12})
  manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
  sd = recipe.recipe_state(manifest, t['seed'])
  net.load_state_dict(sd)
  depthes, confs, rgbs, gt = fusion_inputs(t['B'], t['H'], t['W'], t['seed'] + 1, t['maxdepth'])
  out = {'cfg': np.array([t['maxdepth'], t['B'], t['H'], t['W'], t['seed']] + t['channels'], dtype=np.float64),
         'manifest': np.array(json.dumps([[k, list(s)] for k, s in manifest]))}

  net.train()
  pred = net(depthes, confs, rgbs)
  mask = gt <= t['maxdepth']
  # silog_loss of train_fusion.py:82-88, restated (the script cannot be imported: argparse + dataset loaders at import)
  o, g = torch.squeeze(pred, 1)[mask], gt[mask]
  m2 = (g > 0) * (o > 0)
  d = torch.log(o[m2]) - torch.log(g[m2])
  loss = torch.mean(torch.square(d)) - 0.5 * torch.square(torch.mean(d))
  loss.backward()
  out['train/pred'] = pred.detach().numpy()
  out['train/loss'] = np.array(float(loss.detach()))
  names = [k for k, p in net.named_parameters()]
  out['train/grad_names'] = np.array(names)
  out['train/grad_abs_sum'] = np.array([float(p.grad.double().abs().sum()) for _, p in net.named_parameters()])
  for k, v in net.state_dict().items():  # BatchNorm state after exactly one training forward
    if 'running' in k or 'num_batches' in k:
      out['bn/' + k] = v.numpy().copy()
  net.eval()
  with torch.no_grad():
    out['eval/pred'] = net(depthes, confs, rgbs).numpy()

  # fp64 evaluation of the same network (truth) for the error budget of an fp32 implementation
  net64 = ref.ModeFusion(t['maxdepth'], t['channels'], {'depth': 12, 'rgb': 12}).double()
  net64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
  net64.train()
  d64 = [x.double() for x in depthes], [x.double() for x in confs], [x.double() for x in rgbs]
  out['truth64/train_pred'] = net64(*d64).detach().numpy()
  net64.eval()
  with torch.no_grad():
    out['truth64/eval_pred'] = net64(*d64).numpy()
  np.savez_compressed(os.path.join(HERE, 'fusion_tiny.npz'), **out)
  print('wrote fusion_tiny.npz; train pred range [%.3f, %.3f], loss %.5f, E_ref train %.2e eval %.2e' %
        (out['train/pred'].min(), out['train/pred'].max(), float(loss), np.abs(out['train/pred'] - out['truth64/train_pred']).max(),
         np.abs(out['eval/pred'] - out['truth64/eval_pred']).max()))


if __name__ == '__main__':
  main()
