"""Windowed spherical forward, 128 -> 128 at 256 x 128 on plane-transposed storage: fp32 MFMA kernels against the split-bf16 path."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch
from mode_hip import functional as HF
from models.basic.spherical_conv.sphere_conv import SphereConv
dev = torch.device('cuda', 0)
m = SphereConv(256, 128, 'Cassini', 128, 128, 3, 1, 1).to(dev)
pos = m.position_on(dev)
H, W = pos.shape[2:]
w = m.weight.detach()
for B in (2, 4, 8):
  xt = torch.randn(B, 128, W, H, device=dev)
  yt = torch.empty_like(xt)
  for a in ('f32', 'bf16x6'):
    HF.set_conv_arith(a)
    for _ in range(3):
      HF.sphere_conv_fwd_t(xt, pos, w, yt, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      HF.sphere_conv_fwd_t(xt, pos, w, yt, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('sphere_conv_fwd_t 128->128 %d images %-7s %.3f ms  %.1f TFLOP/s' % (B, a, ms, 2 * 9 * 128 * 128 * B * H * W / ms / 1e9))
