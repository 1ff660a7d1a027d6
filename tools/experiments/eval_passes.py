"""Which activations of an eval forward still take a maximum pass (functional.abs_max), and which entry consumed them.  GPU."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'mode-2022_amd'))
import torch
import models
from mode_hip import functional as HF

dev = 'cuda:0'
log = []
real = HF.abs_max
real_check = HF.check
last = []
HF.abs_max = lambda t: (last.append(tuple(t.shape)), real(t))[1]


def check(rc, name):
  if last:
    log.append((name, last.pop()))
  return real_check(rc, name)


HF.check = check
net = models.ModeDisparity(192, 'Sphere', 1024, 512, 'Cassini').to(dev).eval()
with torch.no_grad():
  net(torch.randn(1, 3, 1024, 512, device=dev), torch.randn(1, 3, 1024, 512, device=dev))
print('ModeDisparity eval forward: %d passes' % len(log))
for n, s in log:
  print('  ', n, s)
