"""The stride-1 3-D layers (32 -> 32) at the benchmark volume and at volumes with another channel-plane pitch: ns per voxel."""
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


for (D, H, W) in ((48, 256, 128), (48, 260, 128), (48, 264, 128), (50, 256, 128), (48, 256, 144)):
  x = torch.randn(2, 32, D, H, W, device=dev)
  w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
  gy = torch.randn_like(x)
  nvox = x.numel() / 32
  t_f = timed(lambda: HF.conv3d_fwd(x, w, 1))
  t_d = timed(lambda: HF.conv3d_bwd_data(gy, w, x.shape, 1))
  t_w = timed(lambda: HF.conv3d_bwd_weight(gy, x, 1))
  print('%3d x %3d x %3d  plane %8d B: fwd %.3f ms = %.3f ns/voxel   bwd-data %.3f ms = %.3f   bwd-weight %.3f ms = %.3f' %
        (D, H, W, D * H * W * 4, t_f, t_f * 1e6 / nvox, t_d, t_d * 1e6 / nvox, t_w, t_w * 1e6 / nvox))
