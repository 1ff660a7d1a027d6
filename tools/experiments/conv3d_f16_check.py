#!/usr/bin/env python3
"""The stride-1 3-D layers on the two-piece fp16 arithmetic (functional.CONV3D_S1_F16) against float64 and against the product arithmetic:
forward and input gradient on unit-variance data, on activations spanning six decades, on gradient-sized data (x 1e-7, the case the
unscaled experiment returned noise for), with the accumulate form; then the bench step with and without it."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'mode-2022_amd'))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from mode_hip import functional as HF

dev = 'cuda:0'
torch.manual_seed(0)


def report(name, got, ref, base):
  err, eb = (got.double().cpu() - ref).abs(), (base.double().cpu() - ref).abs()
  big = ref.abs() > 1e-6 * ref.abs().max()
  print('  %-34s f16x3 max err %.2e (bf16x6 %.2e) of max %.2e; worst relative error over |y| > 1e-6 max: %.2e (bf16x6 %.2e)' % (
      name, float(err.max()), float(eb.max()), float(ref.abs().max()), float((err / ref.abs().clamp_min(1e-300))[big].max()),
      float((eb / ref.abs().clamp_min(1e-300))[big].max())))


for case, mk in (('unit variance', lambda x: x), ('six decades along a row', lambda x: torch.relu(x) * torch.logspace(0, -6, x.shape[-1], device=dev)),
                 ('gradient-sized (x 1e-7)', lambda x: x * 1e-7), ('one outlier x 1e4', None)):
  x = torch.randn(2, 32, 6, 20, 40, device=dev)
  if mk is None:
    x[0, 3, 2, 5, 7] = 1e4
  else:
    x = mk(x)
  w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
  acc = torch.randn_like(x) * float(x.abs().max()) * 0.1
  ref_f = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, 1)
  ref_b = torch.nn.grad.conv3d_input(x.shape, w.double().cpu(), x.double().cpu(), 1, 1)
  gyw = torch.randn_like(x) * (1e-7 if 'gradient' in case else 1.0)
  ref_w = torch.nn.grad.conv3d_weight(x.double().cpu(), w.shape, gyw.double().cpu(), 1, 1)
  print(case)
  res = {}
  for f16 in (False, True):
    HF.CONV3D_S1_F16 = f16
    res[f16] = (HF.conv3d_fwd(x, w, 1), HF.conv3d_bwd_data(x, w, x.shape, 1), HF.conv3d_bwd_data(x, w, x.shape, 1, acc=acc),
                HF.conv3d_bwd_weight(gyw, x, 1))
  report('forward', res[True][0], ref_f, res[False][0])
  report('input gradient', res[True][1], ref_b, res[False][1])
  report('input gradient + acc', res[True][2], ref_b + acc.double().cpu(), res[False][2])
  report('weight gradient', res[True][3], ref_w, res[False][3])
HF.CONV3D_S1_F16 = False
