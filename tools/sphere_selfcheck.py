#!/usr/bin/env python3
"""Fast GPU-side check while iterating on the spherical split kernels: split path against the fp32 kernels on the same inputs
(benchmark shape, 4 images, plane-transposed storage) + HIP-event timings.  The float64 tests are tests/test_gpu_split.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402

from mode_hip import functional as HF  # noqa: E402
from models.basic.spherical_conv.sphere_conv import SphereConv  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = SphereConv(256, 128, 'Cassini', 128, 128, 3, 1, 1).to(dev)
pos = m.position_on(dev)
H, W = pos.shape[2:]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
xt = torch.randn(B, 128, W, H, device=dev)
gyt = torch.randn_like(xt)
w = m.weight.detach()


def t_ms(fn, n=20):
  fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / n


res = {}
for split in (True, False):
  HF.SPHERE_BWD_DATA_SPLIT = split
  HF.SPHERE_BWD_WEIGHT_SPLIT = split
  gx = torch.empty_like(xt)
  gw = torch.zeros_like(w)
  HF.sphere_conv_bwd_data_t(gyt, pos, w, gx, 1)
  HF.sphere_conv_bwd_weight_t(gyt, pos, xt, gw, 1)
  res[split] = (gx, gw, t_ms(lambda: HF.sphere_conv_bwd_data_t(gyt, pos, w, gx, 1)),
                t_ms(lambda: HF.sphere_conv_bwd_weight_t(gyt, pos, xt, torch.zeros_like(w), 1)))
yt = torch.empty_like(xt)
t_f = t_ms(lambda: HF.sphere_conv_fwd_t(xt, pos, w, yt, 1))
(gx1, gw1, td1, tw1), (gx0, gw0, td0, tw0) = res[True], res[False]
print('B=%d  fwd %.3f ms | bwd_data split %.3f ms (fp32 gather %.3f)  max|diff| %.2e of %.2e | bwd_weight split %.3f ms (fp32 %.3f)  max|diff| %.2e of %.2e' %
      (B, t_f, td1, td0, float((gx1 - gx0).abs().max()), float(gx0.abs().max()), tw1, tw0, float((gw1 - gw0).abs().max()), float(gw0.abs().max())))
assert float((gx1 - gx0).abs().max()) < 1e-4 * float(gx0.abs().max()) and float((gw1 - gw0).abs().max()) < 2e-5 * float(gw0.abs().max())
