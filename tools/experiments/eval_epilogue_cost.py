"""Cost of the residual-add epilogue of the eval-mode (folded BatchNorm) split kernels: same layer with and without `add`."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch
from mode_hip import functional as HF
dev = 'cuda:0'


def timeit(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


with torch.no_grad():
  for B in (1, 2):
    x = torch.randn(B, 32, 48, 256, 128, device=dev)
    w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
    bn = torch.nn.BatchNorm3d(32).to(dev).eval()
    add = torch.randn_like(x)
    for a in ('f32', 'bf16x6'):
      HF.set_conv_arith(a)
      t0 = timeit(lambda: HF.conv3d_fwd(x, w, 1))
      t1 = timeit(lambda: HF.conv3d_bn_eval(x, w, bn, 1, None, True))
      t2 = timeit(lambda: HF.conv3d_bn_eval(x, w, bn, 1, add, True))
      print('conv3d 32->32 B=%d %-7s plain %.3f ms | folded BN + ReLU %.3f | + residual add %.3f' % (B, a, t0, t1, t2))
  x = torch.randn(2, 64, 256, 128, device=dev)
  w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
  bn = torch.nn.BatchNorm2d(64).to(dev).eval()
  add = torch.randn_like(x)
  for a in ('f32', 'bf16x6'):
    HF.set_conv_arith(a)
    t0 = timeit(lambda: HF.conv2d_fwd(x, w, 1))
    t1 = timeit(lambda: HF.conv2d_bn_eval(x, w, bn, 1, None, True))
    t2 = timeit(lambda: HF.conv2d_bn_eval(x, w, bn, 1, add, True))
    print('conv2d 64->64 256x128 x2 %-7s plain %.3f ms | folded BN + ReLU %.3f | + residual add %.3f' % (a, t0, t1, t2))
