"""is the stride-2 3-D family bound by DRAM bank conflicts of its power-of-two channel-plane pitch?
The kernels address a contiguous NCDHW tensor, so the plane pitch is changed through the shape: the same layers (32 -> 64 stride-2
forward, its input gradient = the 64 -> 32 transposed convolution, the stride-2 weight gradient) at the benchmark volume 48 x 256 x 128
(plane = 6 MiB = 1.5 M floats) and at volumes whose plane size is NOT a multiple of 4 KiB (W = 132, 136; H = 260), timed per output voxel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch
from mode_hip import functional as HF
dev = torch.device('cuda', 0)
HF.set_conv_arith('bf16x6')


def timed(fn):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / 10


for (D, H, W) in ((48, 256, 128), (48, 256, 132), (48, 256, 136), (48, 260, 128), (48, 256, 144)):
  x = torch.randn(2, 32, D, H, W, device=dev)
  w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
  y = HF.conv3d_fwd(x, w, 2)
  gy = torch.randn_like(y)
  wt = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05  # ConvTranspose3d(64, 32) weight layout (Cin, Cout, 3, 3, 3)
  xs = torch.randn(2, 64, D // 2, H // 2, W // 2, device=dev)
  nvox = y.numel() / 64
  t_f = timed(lambda: HF.conv3d_fwd(x, w, 2))
  t_d = timed(lambda: HF.deconv3d_fwd(xs, wt))
  t_w = timed(lambda: HF.conv3d_bwd_weight(gy, x, 2))
  print('%3d x %3d x %3d  plane %8d B (%% 4096 = %4d): stride-2 fwd %.3f ms = %.2f ns/voxel   transposed %.3f ms = %.2f   stride-2 weight gradient %.3f ms = %.2f' %
        (D, H, W, D * H * W * 4, (D * H * W * 4) % 4096, t_f, t_f * 1e6 / nvox, t_d, t_d * 1e6 / nvox, t_w, t_w * 1e6 / nvox))
