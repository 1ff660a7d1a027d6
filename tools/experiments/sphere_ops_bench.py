"""The three spherical operators at the benchmark shape (128 -> 128, 256 x 128, plane-transposed storage, split-bf16 arithmetic):
HIP-event time per call over 20 back-to-back calls, at 2 / 4 / 8 images."""
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
HF.set_conv_arith('bf16x6')


def timed(fn, what, B):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    fn()
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 20
  print('%-28s %d images  %.3f ms  %.1f TFLOP/s' % (what, B, ms, 2 * 9 * 128 * 128 * B * H * W / ms / 1e9))


for B in (2, 4, 8):
  xt = torch.randn(B, 128, W, H, device=dev)
  yt = torch.empty_like(xt)
  gyt = torch.randn(B, 128, W, H, device=dev)
  gw = torch.zeros_like(w)
  gxt = torch.empty_like(xt)
  timed(lambda: HF.sphere_conv_fwd_t(xt, pos, w, yt, 1), 'sphere_conv_fwd_t', B)
  timed(lambda: HF.sphere_conv_bwd_data_t(gyt, pos, w, gxt, 1), 'sphere_conv_bwd_data_t', B)
  timed(lambda: HF.sphere_conv_bwd_weight_t(gyt, pos, xt, gw, 1), 'sphere_conv_bwd_weight_t', B)
