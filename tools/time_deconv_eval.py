#!/usr/bin/env python3
"""Eval-path transposed convolutions of the hourglass at one pair: plain kernel against the folded-BatchNorm form (with / without residual)."""
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
  ts = []
  for _ in range(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
  return sorted(ts)[n // 2]


with torch.no_grad():
  for (cin, cout, D, H, W) in ((64, 64, 12, 64, 32), (64, 32, 24, 128, 64)):
    x = torch.randn(1, cin, D, H, W, device=dev)
    w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
    bn = torch.nn.BatchNorm3d(cout).to(dev).eval()
    add = torch.randn(1, cout, 2 * D, 2 * H, 2 * W, device=dev)
    r = []
    for arith in ('bf16x6', 'f32'):
      HF.set_conv_arith(arith)
      r.append(t_ms(lambda: HF.deconv3d_fwd(x, w)))
    HF.set_conv_arith('bf16x6')
    print('deconv %d->%d from %dx%dx%d, B=1: plain split %.4f  plain fp32 %.4f | bn_eval %.4f  bn_eval+relu %.4f  bn_eval+add+relu %.4f' % (
        cin, cout, D, H, W, r[0], r[1], t_ms(lambda: HF.deconv3d_bn_eval(x, w, bn, None, False)),
        t_ms(lambda: HF.deconv3d_bn_eval(x, w, bn, None, True)), t_ms(lambda: HF.deconv3d_bn_eval(x, w, bn, add, True))))
