#!/usr/bin/env python3
"""Per-kernel timings at the benchmark shapes (GPU box).  Prints one line per kernel: avg ms, GB/s, TFLOP/s.

  python tools/microbench.py [--batch 2] [--iters 10] [--only cost,sphere,conv3d,head,vendor,stages,export,fusion]
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
  print('%-46s %9.3f ms  %8.1f GB/s  %8.2f TFLOP/s' % (name, ms, nbytes / ms / 1e6, flops / ms / 1e9), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=2)
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--only', default='cost,sphere,conv3d,conv2d,head,vendor,stages,export,fusion')
  a = ap.parse_args()
  only = a.only.split(',')
  dev = 'cuda:0'
  B = a.batch
  torch.manual_seed(0)

  if 'fusion' in only:  # ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12}) at 1024x512
    import models
    from models import stage3d
    sys.path.insert(0, ROOT)
    from oracle import fusion_ref
    net = models.ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12}).to(dev).train()
    depthes = [torch.rand(B, 1, 1024, 512, device=dev) * 50 for _ in range(6)]
    confs = [torch.rand(B, 1, 1024, 512, device=dev) for _ in range(6)]
    rgbs = [torch.rand(B, 3, 1024, 512, device=dev) for _ in range(4)]
    gt = torch.rand(B, 1024, 512, device=dev) * 60

    def fusion_step():
      net.zero_grad(set_to_none=True)
      fusion_ref.training_loss(net(depthes, confs, rgbs), gt, 1000).backward()

    def fusion_eval():
      with torch.no_grad():
        net(depthes, confs, rgbs)

    report('ModeFusion fwd+bwd B=%d (fused BatchNorm kernels)' % B, timeit(fusion_step, max(3, a.iters // 2)))
    net.eval()
    report('ModeFusion eval fwd B=%d' % B, timeit(fusion_eval, max(3, a.iters // 2)))
    net.train()
    fused = stage3d.bn_act
    stage3d.bn_act = stage3d.bn_act_torch  # A/B only: the same module tree on torch's BatchNorm + ReLU
    report('  (same step with torch BatchNorm + ReLU)', timeit(fusion_step, max(3, a.iters // 2)))
    stage3d.bn_act = fused
    del net, depthes, confs, rgbs, gt
    torch.cuda.empty_cache()

  if 'export' in only:  # disparity -> depth -> other camera's view, 1024x512
    import math
    import time
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import geometry_ref as G
    from utils import geometry as HG
    H, W = 1024, 512
    disp = torch.rand(H, W, device=dev) * 40
    disp[torch.rand(H, W, device=dev) < 0.05] = 0
    conf = torch.rand(H, W, device=dev)
    n = H * W
    report('disp2depth 1024x512', timeit(lambda: HG.disp2depth_gpu(disp, conf, '12'), a.iters), 8 * n)
    depth = HG.disp2depth_gpu(disp, conf, '12')[0]
    report('depthViewTransWithConf (project + z-buffer)', timeit(lambda: HG.depthViewTransWithConf_gpu(depth, conf, 0, -1, 0, 0.5 * math.pi, 0, 0),
                                                              a.iters), (4 + 8 + 8 + 4 + 8) * n)
    img = torch.rand(1, 2, H, W, device=dev)
    report('rotateCassini 2 channels', timeit(lambda: HG.rotateCassini_gpu(img, 0.5 * math.pi, 0, 0), a.iters), (8 + 2 * 4 * 2) * n)
    report('disp2depth pair 24 (depth + view transform)', timeit(lambda: HG.disp2depth_gpu(disp, conf, '24'), a.iters), 0)
    dn, cn = disp.cpu().numpy(), conf.cpu().numpy()
    t0 = time.time()
    G.disp2depth(dn, cn, '24')
    print('  (CPU oracle, same call: %.0f ms; numpy maps + interpreted sequential z-buffer -- the reference jit-compiles that loop)' %
          (1e3 * (time.time() - t0)), flush=True)
    t0 = time.time()
    G.project(depth.cpu().numpy(), 0, -1, 0, 0.5 * math.pi, 0, 0)
    print('  (CPU oracle, numpy projection alone, what the reference runs on the host before its jitted loop: %.0f ms)' %
          (1e3 * (time.time() - t0)), flush=True)

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

  if 'conv3d' in only or 'conv3d_main' in only:
    shapes = ((64, 32, 48, 256, 128, 1), (32, 32, 48, 256, 128, 1), (32, 64, 48, 256, 128, 2), (64, 64, 24, 128, 64, 1),
              (64, 64, 24, 128, 64, 2), (64, 64, 12, 64, 32, 1), (32, 1, 48, 256, 128, 1))
    if 'conv3d_main' in only:  # the single dominant layer shape (used for the PMC traffic passes)
      shapes = ((32, 32, 48, 256, 128, 1),)
    for (ci, co, d, h, w_, s) in shapes:
      x = torch.randn(B, ci, d, h, w_, device=dev)
      wt = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
      y = HF.conv3d_fwd(x, wt, s)
      fl = 2 * y.numel() * ci * 27
      nb = 4 * (x.numel() + y.numel())
      tag = '%d->%d s%d @%dx%dx%d' % (ci, co, s, d, h, w_)
      report('conv3d_fwd ' + tag, timeit(lambda: HF.conv3d_fwd(x, wt, s), a.iters), nb, fl)
      report('  (vendor conv3d fwd)', timeit(lambda: F.conv3d(x, wt, None, s, 1), max(2, a.iters // 3)), nb, fl)
      gy = torch.randn_like(y)
      report('conv3d_bwd_data ' + tag, timeit(lambda: HF.conv3d_bwd_data(gy, wt, x.shape, s), a.iters), nb, fl)
      report('conv3d_bwd_weight ' + tag, timeit(lambda: HF.conv3d_bwd_weight(gy, x, s), a.iters), nb, fl)
      del x, wt, y, gy
    for (ci, co, d, h, w_) in ((64, 64, 12, 64, 32), (64, 32, 24, 128, 64)):
      x = torch.randn(B, ci, d, h, w_, device=dev)
      wt = torch.randn(ci, co, 3, 3, 3, device=dev) * 0.05
      y = HF.deconv3d_fwd(x, wt)
      fl = 2 * x.numel() * co * 27
      nb = 4 * (x.numel() + y.numel())
      report('deconv3d_fwd %d->%d @%dx%dx%d' % (ci, co, d, h, w_), timeit(lambda: HF.deconv3d_fwd(x, wt), a.iters), nb, fl)
      report('  (vendor conv_transpose3d fwd)', timeit(lambda: F.conv_transpose3d(x, wt, None, 2, 1, 1), max(2, a.iters // 3)), nb, fl)
      del x, wt, y

  if 'conv2d' in only:
    # regular 3x3 layers of the extractor at the step's 4 images (paired pass): own kernels vs the vendor library
    B4 = 2 * B
    for (ci, co, h, w_, dil) in ((64, 64, 256, 128, 1), (64, 64, 256, 128, 2), (128, 128, 256, 128, 1), (64, 64, 512, 256, 1), (32, 32, 512, 256, 1)):
      x = torch.randn(B4, ci, h, w_, device=dev)
      wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
      gy = torch.randn(B4, co, h, w_, device=dev)
      fl = 2 * gy.numel() * ci * 9
      nb = 4 * (x.numel() + gy.numel())
      tag = '%d->%d d%d @%dx%d B=%d' % (ci, co, dil, h, w_, B4)
      report('conv2d_fwd ' + tag, timeit(lambda: HF.conv2d_fwd(x, wt, dil), a.iters), nb, fl)
      report('  (vendor conv2d fwd)', timeit(lambda: F.conv2d(x, wt, None, 1, dil, dil), a.iters), nb, fl)
      report('conv2d_bwd_data ' + tag, timeit(lambda: HF.conv2d_bwd_data(gy, wt, dil), a.iters), nb, fl)
      report('  (vendor conv2d input gradient)', timeit(lambda: torch.ops.aten.convolution_backward(
          gy, x, wt, None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [True, False, False]), a.iters), nb, fl)
      report('conv2d_bwd_weight ' + tag, timeit(lambda: HF.conv2d_bwd_weight(gy, x, dil), a.iters), nb, fl)
      report('  (vendor conv2d weight gradient)', timeit(lambda: torch.ops.aten.convolution_backward(
          gy, x, wt, None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [False, True, False]), a.iters), nb, fl)
      del x, wt, gy

  if 'head' in only:
    lg = torch.randn(B, 1, 48, 256, 128, device=dev)
    nb = 4 * (lg.numel() + B * 1024 * 512)
    report('head_fwd', timeit(lambda: HF.head_fwd(lg, (192, 1024, 512)), a.iters), nb)
    report('head_fwd + confidence', timeit(lambda: HF.head_fwd(lg, (192, 1024, 512), True), a.iters), nb)
    g = torch.randn(B, 1, 1024, 512, device=dev)
    report('head_bwd', timeit(lambda: HF.head_bwd(lg, g, (192, 1024, 512)), a.iters), nb + 4 * lg.numel())
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import plain_ops
    report('  (vendor upsample+softmax+regress fwd)', timeit(lambda: plain_ops.head(lg, (192, 1024, 512)), 3), nb)

  if 'vendor' in only:
    x = torch.randn(B, 32, 48, 256, 128, device=dev)
    bn = torch.nn.BatchNorm3d(32).to(dev)
    report('vendor BatchNorm3d train 32ch fwd', timeit(lambda: bn(x), a.iters), 2 * 4 * x.numel())
    xr = x.clone().requires_grad_(True)

    def bnb():
      xr.grad = None
      bn(xr).sum().backward()

    report('vendor BatchNorm3d fwd+bwd', timeit(bnb, 3), 5 * 4 * x.numel())
    report('vendor relu', timeit(lambda: F.relu(x), a.iters), 2 * 4 * x.numel())
    report('vendor add', timeit(lambda: x + x, a.iters), 3 * 4 * x.numel())
    x2 = torch.randn(B, 64, 512, 256, device=dev)
    bn2 = torch.nn.BatchNorm2d(64).to(dev)
    report('vendor BatchNorm2d train 64ch @512x256', timeit(lambda: bn2(x2), a.iters), 2 * 4 * x2.numel())

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
      from models import stage3d
      c0 = stage3d.conv_bn(net.dres0[0], cost, relu=True)
      report('stage: hourglass fwd', timeit(lambda: net.dres2(c0, None, None), 3, 1))
      report('stage: classif fwd', timeit(lambda: stage3d.classify(net.classif1, c0), 3, 1))
      report('stage: full forward (train mode, 3 heads)', timeit(lambda: net(left, right), 3, 1))

    def fe():
      net.zero_grad()
      net.feature_extraction(left).sum().backward()

    report('stage: feature_extraction fwd+bwd', timeit(fe, 3, 1))

    def full():
      net.zero_grad()
      o = net(left, right)
      (o[0].mean() + o[1].mean() + o[2].mean()).backward()

    report('stage: full fwd+bwd', timeit(full, 3, 1))
    print('peak memory GB', torch.cuda.max_memory_allocated() / 2**30)


if __name__ == '__main__':
  main()
