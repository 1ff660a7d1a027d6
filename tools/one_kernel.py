#!/usr/bin/env python3
"""Runs ONE kernel shape a few times, for PMC passes that must not mix shapes (rocprofv3 --pmc ... -- python3 tools/one_kernel.py <what>).
  conv3d_bwd_weight_32   : mode_conv3d_bwd_weight 32->32 at 48x256x128, batch 2 (the bench's roofline kernel)
  conv3d_fwd_32          : mode_conv3d_fwd 32->32, same volume
  bn3d_32                : BatchNorm3d(32) + ReLU training forward + backward on the same volume (mode_bn_train_fwd / _bwd)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402

from mode_hip import functional as HF  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'conv3d_bwd_weight_32'
dev = torch.device('cuda', 0)
x = torch.randn(2, 32, 48, 256, 128, device=dev)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
gy = torch.randn_like(x)
if what == 'bn3d_32':
  bn = torch.nn.BatchNorm3d(32).to(dev)
  xr = x.clone().requires_grad_(True)
  for _ in range(4):
    xr.grad = None
    out = HF.bn_act(bn, xr, None, True)
    out.backward(gy)
  torch.cuda.synchronize()
  print('done', what)
  sys.exit(0)
for _ in range(4):
  if what == 'conv3d_bwd_weight_32':
    HF.conv3d_bwd_weight(gy, x, 1)
  else:
    HF.conv3d_fwd(x, w, 1)
torch.cuda.synchronize()
print('done', what)
