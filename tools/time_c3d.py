#!/usr/bin/env python3
"""Times the stride-1 3-D convolution kernels of the regulariser in isolation (benchmark shapes, B = 2), 20 launches each after
warm-up, torch events on the current stream (HF launches on it): ms per launch.  For A/B runs of one kernel change on one box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402

from mode_hip import functional as HF  # noqa: E402

dev = torch.device('cuda', 0)


def t_ms(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / n


out = []
if 's2w' in sys.argv:
  # stride-2 / transposed weight gradients of the hourglass: (x channels, gy channels, x volume), B = 2, both arithmetics
  shapes = [(32, 64, 48, 256, 128), (64, 64, 24, 128, 64), (64, 64, 12, 64, 32)]
  if 'strides' in sys.argv:  # does the time depend on the channel stride (48 x 256 x 128 x 4 B = 3 * 2^21: every channel on the same DRAM bank bits)?
    shapes = [(32, 64, 48, 256, 128), (32, 64, 50, 256, 128), (32, 64, 48, 264, 128), (32, 64, 46, 256, 128)]
  for (ci, co, D, H, W) in shapes:
    x = torch.randn(2, ci, D, H, W, device=dev)
    gy = torch.randn(2, co, D // 2, H // 2, W // 2, device=dev)
    r = []
    for arith in ('bf16x6', 'f32'):
      HF.set_conv_arith(arith)
      r.append(t_ms(lambda: HF.conv3d_bwd_weight(gy, x, 2)))
    HF.set_conv_arith('bf16x6')
    out.append('s2 bwd_weight x%d gy%d %dx%dx%d: split %.4f  fp32 %.4f' % (ci, co, D, H, W, r[0], r[1]))
  print(' | '.join(out))
  sys.exit(0)
vols = ((32, 48, 256, 128), (64, 24, 128, 64), (64, 12, 64, 32))
if 'strides' in sys.argv:  # per-voxel time against the channel stride (see the s2w case)
  vols = ((32, 48, 256, 128), (32, 50, 256, 128), (32, 46, 256, 128), (32, 48, 264, 128), (32, 48, 256, 136))
for (C, D, H, W) in vols:
  x = torch.randn(2, C, D, H, W, device=dev)
  w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
  gy = torch.randn_like(x)
  gw = torch.zeros_like(w)
  bn = torch.nn.BatchNorm3d(C).to(dev)
  xr = x.clone().requires_grad_(True)

  def bn_step():
    xr.grad = None
    HF.bn_act(bn, xr, None, True).backward(gy)

  ts = (t_ms(lambda: HF.conv3d_fwd(x, w, 1)), t_ms(lambda: HF.conv3d_bwd_data(gy, w, x.shape, 1)), t_ms(lambda: HF.conv3d_bwd_weight(gy, x, 1)),
        t_ms(bn_step))
  nv = D * H * W / (48 * 256 * 128)
  out.append('%d->%d %dx%dx%d: fwd %.4f  bwd_data %.4f  bwd_weight %.4f  bn fwd+bwd %.4f   [per 48x256x128 voxels: %.4f %.4f %.4f %.4f]' % (
      (C, C, D, H, W) + ts + tuple(t / nv for t in ts)))
print(' | '.join(out))
