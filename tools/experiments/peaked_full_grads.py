"""Per-tensor gradient errors of the HIP path on the full-size peaked fixture, both arithmetics (which tensors are furthest from the
reference, and how far the reference itself is from float64).   python tools/experiments/peaked_full_grads.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
  sys.path.insert(0, p)
import numpy as np
import torch
import recipe
from oracle import mode_ref
import test_gpu_parity as T
from mode_hip import functional as HF

z = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_peaked_full.npz'))
own = z['truth64/grad_rel_l2']
names = [str(n) for n in z['train/grad_names']]
for arith in ('f32', 'bf16x6'):
  HF.set_conv_arith(arith)
  net, left, right, gt, seed = T._load(z)
  net.train()
  preds = net(left, right)
  loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
  loss.backward()
  grads = dict(net.named_parameters())
  K = z['train/grad_proj'].shape[1]
  rows = []
  for i, name in enumerate(names):
    g = grads[name].grad.detach().cpu().numpy().astype(np.float64).reshape(-1)
    proj = recipe.projection_signs(seed, i, g.size, K).astype(np.float64) @ g
    d = proj - z['train/grad_proj'][i]
    norm = float(z['train/grad_norm'][i])
    rel = float(np.sqrt(np.mean(d ** 2)) / max(norm, 1e-30))
    rows.append((rel / max(own[i], 1e-9), rel, own[i], name))
  rows.sort(reverse=True)
  print('== %s: tensors furthest from the reference, relative to the reference\'s own fp32-vs-fp64 error' % arith)
  for r in rows[:12]:
    print('   %-52s rel %.2e   reference own %.2e   ratio %.1f' % (r[3], r[1], r[2], r[0]))
