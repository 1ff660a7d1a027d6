"""Development aid: run an eval forward with the folded BatchNorm path and, layer by layer, compare with the two-step form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests', 'golden')):
  sys.path.insert(0, p)
import torch
import recipe
import models
from models import stage3d
from mode_hip import functional as HF
DEV = 'cuda:0'
net = models.ModeDisparity(32, 'Sphere', 128, 64, 'Cassini').to(DEV)
net.load_state_dict(recipe.recipe_state_wc(recipe.load_manifest(), 77))
left, right = [t.to(DEV) for t in recipe.recipe_images(2, 128, 64, 78)]
bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
for m in bns:
  m.momentum = 1.0
net.train()
with torch.no_grad():
  net(left, right)
net.eval()
orig = stage3d._conv_bn_folded
names = {id(m): n for n, m in net.named_modules()}
def checked(conv, bn, x, add, relu):
  y = orig(conv, bn, x, add, relu)
  ref = stage3d.bn_act(bn, stage3d.conv3(conv, x), add, relu)
  if y is not None:
    err = float((y - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    flag = '' if err < 1e-5 else '   <<<<<<'
    print('%-45s %-14s in %s out %s  nan(in) %d nan(folded) %d nan(two-step) %d  err %.3e%s' % (
        names.get(id(conv), '?'), type(conv).__name__, tuple(x.shape), tuple(y.shape), int(torch.isnan(x).sum()), int(torch.isnan(y).sum()),
        int(torch.isnan(ref).sum()), err, flag))
  return ref  # continue on the two-step values so that one bad layer does not mask the next
stage3d._conv_bn_folded = checked
with torch.no_grad():
  out = net(left, right)
print('final nan', int(torch.isnan(out).sum()))
