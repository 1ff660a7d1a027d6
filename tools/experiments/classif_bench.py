#!/usr/bin/env python3
"""Times the classifier head of the training step at the benchmark volume (2 x 32 x 48 x 256 x 128): the fused operator
(functional.ClassifHeadFunction, csrc/classif_head.hip) against the composition of separate operators it replaces, forward and backward,
HIP events around each half, with a 1 GiB fill between repetitions so that nothing is served from the Infinity Cache.
    python tools/experiments/classif_bench.py            (under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch
import torch.nn as nn
from mode_hip import functional as HF

dev = 'cuda:0'
B, C, D, H, W = 2, 32, 48, 256, 128
torch.manual_seed(0)
y = (torch.randn(B, C, D, H, W, device=dev) * 1.3 + 0.2).requires_grad_(True)
add = torch.randn(B, 1, D, H, W, device=dev)
go = torch.randn(B, 1, D, H, W, device=dev)
bn = nn.BatchNorm3d(C).to(dev).train()
conv = nn.Conv3d(C, 1, 3, padding=1, bias=False).to(dev)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def timed(fn, reps=5):
  ts = []
  for _ in range(reps):
    flush.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
  return min(ts), out


VARIANTS = (('fused', lambda: HF.classif_head_train(y, bn, conv, add)),
            ('unfused', lambda: HF.conv3d(HF.bn_act(bn, y, None, True), conv.weight, 1) + add))
for name, fwd in VARIANTS:
  if len(sys.argv) > 1 and name not in sys.argv[1:]:
    continue
  for _ in range(2):
    fwd().backward(go)
  y.grad = None
  tf, cost = timed(fwd)
  tb, _ = timed(lambda: torch.autograd.grad(cost, (y, conv.weight, bn.weight, bn.bias), go, retain_graph=True))
  print('%-8s forward %.3f ms   backward %.3f ms' % (name, tf, tb))
