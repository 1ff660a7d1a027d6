#!/usr/bin/env python3
"""Times the spherical operator of the extractor's layer4 in isolation (128 -> 128 at 256 x 128 Cassini, 4 images, plane-transposed
storage as in the network): forward, input gradient, weight gradient, ms per call (median of 20)."""
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


from models.basic.spherical_conv.sphere_conv import SphereConv  # noqa: E402

m = SphereConv(256, 128, 'Cassini', 128, 128, 3, 1, 1).to(dev)
pos = m.position_on(dev)
H, W = pos.shape[2:]
xt = torch.randn(4, 128, W, H, device=dev)
gyt = torch.randn_like(xt)
w = m.weight.detach()
yt = torch.empty_like(xt)
gxt = torch.empty_like(xt)
gw = torch.zeros_like(w)
if 'check' in sys.argv:  # the split forward against the fp32 windowed kernels on the same inputs
  y1 = torch.empty_like(xt)
  HF.sphere_conv_fwd_t(xt, pos, w, y1, 1)
  HF.set_conv_arith('f32')
  y0 = torch.empty_like(xt)
  HF.sphere_conv_fwd_t(xt, pos, w, y0, 1)
  HF.set_conv_arith('bf16x6')
  e, sc = float((y1 - y0).abs().max()), float(y0.abs().max())
  print('fwd split vs fp32 kernels: max|diff| %.2e of %.2e' % (e, sc))
  assert e < 1e-4 * sc
for arith in (['bf16x6', 'f32'] if 'both' in sys.argv else ['bf16x6']):
  HF.set_conv_arith(arith)
  out = ['fwd %.4f' % t_ms(lambda: HF.sphere_conv_fwd_t(xt, pos, w, yt, 1))]
  if 'fwd' not in sys.argv:
    out += ['bwd_data %.4f' % t_ms(lambda: HF.sphere_conv_bwd_data_t(gyt, pos, w, gxt, 1)),
            'bwd_weight %.4f' % t_ms(lambda: HF.sphere_conv_bwd_weight_t(gyt, pos, xt, gw, 1))]
  print('sphere 128->128 256x128 x4 [%s]: ' % arith + '  '.join(out))
