#!/usr/bin/env python3
"""Which vendor 2D convolution of the feature extractor is slow?  Times every distinct Conv2d config fwd / bwd on the GPU.

  python tools/conv2d_probe.py [--benchmark]      (--benchmark sets torch.backends.cudnn.benchmark = True)
"""
import sys

import torch
import torch.nn.functional as F

CONFIGS = [  # (Ci, Co, k, stride, pad, dil, H, W, needs_input_grad)
    (3, 32, 7, 2, 3, 1, 1024, 512, False),
    (32, 32, 3, 1, 1, 1, 512, 256, True),
    (32, 64, 3, 1, 1, 1, 512, 256, True),
    (64, 64, 3, 1, 1, 1, 512, 256, True),
    (32, 64, 1, 1, 0, 1, 512, 256, True),
    (64, 64, 3, 2, 1, 1, 512, 256, True),
    (64, 64, 1, 2, 0, 1, 512, 256, True),
    (64, 64, 3, 1, 1, 1, 256, 128, True),
    (64, 64, 3, 1, 2, 2, 256, 128, True),
    (64, 128, 1, 1, 0, 1, 256, 128, True),
    (256, 128, 1, 1, 0, 1, 256, 128, True),
    (128, 128, 3, 1, 1, 1, 256, 128, True),
    (128, 32, 1, 1, 0, 1, 256, 128, True),
]


def timeit(fn, iters=5, warm=2):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters


def main():
  if '--benchmark' in sys.argv:
    torch.backends.cudnn.benchmark = True
  print('cudnn.benchmark =', torch.backends.cudnn.benchmark)
  B = 2
  for (ci, co, k, s, p, d, H, W, need) in CONFIGS:
    x = torch.randn(B, ci, H, W, device='cuda', requires_grad=need)
    w = torch.randn(co, ci, k, k, device='cuda', requires_grad=True)
    y = F.conv2d(x, w, None, s, p, d)
    gy = torch.randn_like(y)
    fl = 2 * y.numel() * ci * k * k
    tf = timeit(lambda: F.conv2d(x, w, None, s, p, d))

    def fb():
      x.grad = w.grad = None
      F.conv2d(x, w, None, s, p, d).backward(gy)

    tb = timeit(fb, 3, 2)
    print('%3d->%3d k%d s%d d%d @%4dx%3d  fwd %8.3f ms (%6.2f TF)   fwd+bwd %8.3f ms' % (ci, co, k, s, d, H, W, tf, fl / tf / 1e9, tb),
          flush=True)


if __name__ == '__main__':
  main()
