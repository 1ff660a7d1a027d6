#!/usr/bin/env python3
"""Per-kernel timings at the benchmark shapes (GPU box).  Prints one line per kernel: avg ms, GB/s, TFLOP/s.

  python tools/microbench.py [--batch 2] [--iters 10] [--only cost,sphere,vendor3d,stages]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mode_hip import functional as HF  # noqa: E402


def timeit(fn, iters, warm=2):
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


def report(name, ms, nbytes=0, flops=0):
  print('%-38s %9.3f ms  %8.1f GB/s  %8.2f TFLOP/s' % (name, ms, nbytes / ms / 1e6, flops / ms / 1e9), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=2)
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--only', default='cost,sphere,vendor3d,stages')
  a = ap.parse_args()
  only = a.only.split(',')
  dev = 'cuda:0'
  B = a.batch
  torch.manual_seed(0)

  if 'cost' in only:
    C, D4, H, W = 32, 48, 256, 128
    ref, tgt = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
    nb = 4 * (2 * ref.numel() + B * 2 * C * D4 * H * W)
    report('cost_volume_fwd B=%d' % B, timeit(lambda: HF.cost_volume_fwd(ref, tgt, D4), a.iters), nb)
    g = torch.randn(B, 2 * C, D4, H, W, device=dev)
    report('cost_volume_bwd B=%d' % B, timeit(lambda: HF.cost_volume_bwd(g, C), a.iters), nb)
    out = torch.empty_like(g)
    report('  (torch copy_ of the same volume)', timeit(lambda: out.copy_(g), a.iters), 2 * 4 * g.numel())
    del g, out

  if 'sphere' in only:
    from models.basic.spherical_conv.sphere_conv import SphereConv
    for ci, co in ((128, 128), (64, 128)):
      m = SphereConv(256, 128, 'Cassini', ci, co, 3, 1, 1).to(dev)
      pos = m.position_on(torch.device(dev))
      x = torch.randn(B, ci, 256, 128, device=dev)
      w = m.weight.detach()
      y = torch.empty(B, co, 256, 128, device=dev)
      fl = 2 * y.numel() * ci * 9
      nb = 4 * (x.numel() + y.numel() + pos.numel() + w.numel())
      report('sphere_conv_fwd %d->%d B=%d' % (ci, co, B), timeit(lambda: HF.sphere_conv_fwd(x, pos, w, y, (1, 1), 1), a.iters), nb, fl)
      gy = torch.randn_like(y)
      gx = torch.zeros_like(x)
      report('sphere_conv_bwd_data %d->%d' % (ci, co), timeit(lambda: HF.sphere_conv_bwd_data(gy, pos, w, gx, (1, 1), 1), a.iters), nb, fl)
      gw = torch.zeros_like(w)
      report('sphere_conv_bwd_weight %d->%d' % (ci, co), timeit(lambda: HF.sphere_conv_bwd_weight(gy, pos, x, gw, (1, 1), 1), a.iters), nb, fl)
      wr = torch.randn(co, ci, 3, 3, device=dev)
      report('  (vendor conv2d 3x3 same shape)', timeit(lambda: F.conv2d(x, wr, None, 1, 1), a.iters), nb, fl)

  if 'vendor3d' in only:
    for (ci, co, d, h, w_, s) in ((64, 32, 48, 256, 128, 1), (32, 32, 48, 256, 128, 1), (32, 64, 48, 256, 128, 2), (64, 64, 24, 128, 64, 1),
                                  (64, 64, 24, 128, 64, 2), (64, 64, 12, 64, 32, 1), (32, 1, 48, 256, 128, 1)):
      x = torch.randn(B, ci, d, h, w_, device=dev, requires_grad=True)
      wt = torch.randn(co, ci, 3, 3, 3, device=dev, requires_grad=True)
      y = F.conv3d(x, wt, None, s, 1)
      fl = 2 * y.numel() * ci * 27
      nb = 4 * (x.numel() + y.numel())
      report('vendor conv3d %d->%d s%d @%dx%dx%d fwd' % (ci, co, s, d, h, w_), timeit(lambda: F.conv3d(x, wt, None, s, 1), a.iters), nb, fl)
      gy = torch.randn_like(y)

      def bwd():
        x.grad = wt.grad = None
        F.conv3d(x, wt, None, s, 1).backward(gy)

      report('   fwd+bwd', timeit(bwd, max(2, a.iters // 2)), 3 * nb, 3 * fl)
      del x, wt, y, gy
    x = torch.randn(B, 64, 12, 64, 32, device=dev)
    wt = torch.randn(64, 64, 3, 3, 3, device=dev)
    y = F.conv_transpose3d(x, wt, None, 2, 1, 1)
    report('vendor deconv3d 64->64 @12x64x32', timeit(lambda: F.conv_transpose3d(x, wt, None, 2, 1, 1), a.iters), 4 * (x.numel() + y.numel()),
           2 * x.numel() * 64 * 27)
    x = torch.randn(B, 32, 48, 256, 128, device=dev)
    bn = torch.nn.BatchNorm3d(32).to(dev)
    report('vendor BatchNorm3d train 32ch', timeit(lambda: bn(x), a.iters), 2 * 4 * x.numel())
    report('vendor relu', timeit(lambda: F.relu(x), a.iters), 2 * 4 * x.numel())
    lg = torch.randn(B, 1, 48, 256, 128, device=dev)

    def head():
      up = F.interpolate(lg, [192, 1024, 512], mode='trilinear', align_corners=True).squeeze(1)
      p = F.softmax(up, 1)
      return (p * torch.arange(192, device=dev, dtype=p.dtype).view(1, 192, 1, 1)).sum(1, keepdim=True)

    report('vendor head (upsample+softmax+regress)', timeit(head, a.iters), 4 * (lg.numel() + B * 1024 * 512))

  if 'stages' in only:
    import models
    net = models.ModeDisparity(192, 'Sphere', 1024, 512, 'Cassini').to(dev)
    net.train()
    left, right = torch.randn(B, 3, 1024, 512, device=dev), torch.randn(B, 3, 1024, 512, device=dev)
    with torch.no_grad():
      report('stage: feature_extraction fwd (1 image batch)', timeit(lambda: net.feature_extraction(left), 3, 1))
      fea = net.feature_extraction(left)
      cost = HF.cost_volume_fwd(fea, fea, 48)
      report('stage: dres0 fwd', timeit(lambda: net.dres0(cost), 3, 1))
      c0 = net.dres0(cost)
      report('stage: hourglass fwd', timeit(lambda: net.dres2(c0, None, None), 3, 1))
      report('stage: classif fwd', timeit(lambda: net.classif1(c0), 3, 1))
      report('stage: full forward (train mode, 3 heads)', timeit(lambda: net(left, right), 3, 1))

    def full():
      net.zero_grad()
      o = net(left, right)
      (o[0].mean() + o[1].mean() + o[2].mean()).backward()

    report('stage: full fwd+bwd', timeit(full, 3, 1))
    print('peak memory GB', torch.cuda.max_memory_allocated() / 2**30)


if __name__ == '__main__':
  main()
