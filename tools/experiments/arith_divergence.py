"""Where do the two arithmetics ('f32' / 'bf16x6') of the HIP path part ways?  Runs the same training step of a fixture under both with
every operator output sampled (norm + 4096 strided elements) and lists, in execution order, the entries whose samples differ by more
than `tol` relative to the entry's rms.   python tools/experiments/arith_divergence.py [fixture.npz] [tol]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
  sys.path.insert(0, p)
import numpy as np
import torch
from oracle import mode_ref
import op_trace
import test_gpu_parity as T
from mode_hip import functional as HF

fixture = sys.argv[1] if len(sys.argv) > 1 else 'model_peaked_full.npz'
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4
z = np.load(os.path.join(ROOT, 'tests', 'golden', fixture))


class Sampler(op_trace.Trace):
  def record(self, label, t):
    if not (torch.is_tensor(t) and t.is_cuda and t.is_floating_point() and t.numel() > 0):
      return
    self.labels.append('%s %s' % (label, tuple(t.shape)))
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // 4096)
    mask = (t.detach() > 0) if ('BnActFunction.forward' in label and t.dim() == 5) else None  # ReLU masks of the 3-D stage, kept on the GPU
    self.items.append((f[::step][:4096].double().cpu(), float(f.double().pow(2).mean().sqrt()), mask))


runs = {}
for arith in ('f32', 'bf16x6'):
  HF.set_conv_arith(arith)
  net, left, right, gt, seed = T._load(z)
  net.train()
  tr = Sampler()
  with op_trace.tracing(tr):
    preds = net(left, right)
    loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
    loss.backward()
  torch.cuda.synchronize()
  runs[arith] = tr
a, b = runs['f32'], runs['bf16x6']
assert a.labels == b.labels
print('%d entries; listing those whose sampled elements differ by more than %.1e of the entry rms' % (len(a.labels), tol))
flips = []
for lab, (sa, ra, ma), (sb, rb, mb) in zip(a.labels, a.items, b.items):
  if ma is not None:
    flips.append((int((ma != mb).sum()), ma.numel(), lab))
  d = float((sa - sb).abs().max()) / max(ra, 1e-300)
  rms = float((sa - sb).pow(2).mean().sqrt()) / max(ra, 1e-300)
  if d > tol:
    print('   %-70s max %.2e rms %.2e of the entry rms %.3e' % (lab, d, rms, ra))

print('ReLU-mask elements that differ between the two arithmetics, per BatchNorm(+ReLU) output of the 3-D stage:')
for n, tot, lab in flips:
  print('   %6d of %10d  %s' % (n, tot, lab))
