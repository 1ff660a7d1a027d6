"""GPU (-m gpu): a K-step TRAINING TRAJECTORY against the imported reference.

tests/golden/model_steps_tiny.npz holds three iterations of the reference's own loop body (train_disparity.py:147-161: zero_grad, forward,
masked smooth-L1 0.5 / 0.7 / 1.0, backward, Adam(lr 1e-3) step) at 64 x 32 / 16, batch 2.  The product runs the same three iterations
(a) the way bench.py does -- GradAllReducer with gradient sinks, ModeDisparity.forward_loss, zero-grad + forward + loss + backward replayed as
one hipGraph, fused Adam --
and (b) eagerly with plain autograd accumulation and torch's ordinary Adam; both must reproduce the reference's losses, its BatchNorm
running statistics after step 3 and its parameter UPDATE p_3 - p_0.

What can be asked of a trajectory: Adam's first steps move every entry by ~lr * sign(g) (m / sqrt(v) = +-1 at step 1) whatever |g|, so where a
gradient is within fp32 round-off of zero the direction is decided by that round-off -- the reference's own fp32 run and a float64
evaluation of the same three steps differ by 1.6e-4 in the third loss, 6e-3 in the running statistics and 2.5e-2 (median per tensor) in the
update.  The fixture stores that float64 trajectory, and every bound below is max(floor, 2-3 x the reference's own |fp32 - fp64|)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import recipe  # noqa: E402

import models  # noqa: E402
import mode_hip  # noqa: E402
from mode_hip import data_parallel  # noqa: E402
from mode_hip.graph_step import GraphedStep  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LR = 1e-3


def _setup(z):
  maxdisp, H, W, B, seed, K = [int(v) for v in z['cfg']]
  net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini').to(DEV)
  net.load_state_dict(recipe.recipe_state_wc(recipe.load_manifest(), seed))
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity_smooth(B, H, W, seed + 2, maxdisp)
  return net, left.to(DEV), right.to(DEV), gt.to(DEV), K


def _loss(net, left, right, gt, count):
  mask = ~torch.isnan(gt)
  gt0 = torch.nan_to_num(gt)
  o1, o2, o3 = net(left, right)
  loss = 0
  for wgt, o in ((0.5, o1), (0.7, o2), (1.0, o3)):
    loss = loss + wgt * data_parallel.global_masked_mean(F.smooth_l1_loss(o, gt0, reduction='none'), mask, count=count)
  return loss


def _check(z, net, p0, losses, tag):
  """Every quantity is compared with the reference's fp32 trajectory and bounded by max(floor, 3 x E_ref), E_ref = the reference's OWN
  |fp32 - fp64| for that quantity (loss64 / bn64 / dproj64 in the fixture: the float64 oracle stepped by the same optimizer)."""
  ref, ref64 = z['loss'], z['loss64']
  print('%s: losses %s   reference %s   reference in float64 %s' % (tag, ['%.7f' % v for v in losses], ['%.7f' % v for v in ref], ['%.7f' % v for v in ref64]))
  for k, (got, want, w64) in enumerate(zip(losses, ref, ref64)):
    bound = max(2e-5 * abs(want), 3 * abs(want - w64))
    assert abs(got - want) <= bound, (tag, k, got, want, bound)
  sd = net.state_dict()
  worst = e_ref = 0.0
  for key in z.files:
    if key.startswith('bn/'):
      got, want, w64 = sd[key[3:]].detach().cpu().double().numpy(), z[key].astype(np.float64), z['bn64/' + key[3:]]
      scale = max(1.0, np.abs(want).max())
      worst = max(worst, float(np.abs(got - want).max() / scale))
      e_ref = max(e_ref, float(np.abs(want - w64).max() / scale))
  print('%s: BatchNorm running statistics after step %d: worst |diff| / max(1, scale) = %.3e   (reference fp32 vs fp64: %.3e)' %
        (tag, len(losses), worst, e_ref))
  assert worst <= max(2e-4, 3 * e_ref), (tag, worst, e_ref)
  nbt3, nbt2 = [int(v) for v in z['nbt']]
  for key, v in sd.items():
    if key.endswith('num_batches_tracked'):
      assert int(v) == (nbt2 if key.startswith('feature_extraction') else nbt3), key
  names = [str(n) for n in z['names']]
  params = dict(net.named_parameters())
  rel, rel_ref, outliers, entries = [], [], 0, 0
  seed = int(z['cfg'][4])
  for i, name in enumerate(names):
    d = (params[name].detach().cpu().double() - p0[name]).reshape(-1).numpy()
    proj = recipe.projection_signs(seed, i, d.size, z['dproj'].shape[1]).astype(np.float64) @ d
    # relative L2 of the update from its projections (an unbiased estimate of ||d - d_ref||^2 / ||d_ref||^2) ...
    den = max(z['dnorm'][i]**2, 1e-30)
    rel.append((float(((proj - z['dproj'][i])**2).mean()) / den)**0.5)
    rel_ref.append((float(((z['dproj64'][i] - z['dproj'][i])**2).mean()) / den)**0.5)
    # ... and the sampled entries
    got = params[name].detach().cpu().double().reshape(-1).numpy()[z['idx'][i]]
    bad = np.abs(got - z['pK_val'][i]) > 3 * LR
    outliers += int(bad.sum())
    entries += bad.size
  rel, rel_ref = np.array(rel), np.array(rel_ref)
  print('%s: relative L2 of the parameter update per tensor: median %.3e, 90th percentile %.3e, max %.3e (%s);  the reference fp32 vs fp64: '
        'median %.3e, 90th %.3e, max %.3e;  sampled entries off by > 3 lr: %d of %d' %
        (tag, np.median(rel), np.percentile(rel, 90), rel.max(), names[int(rel.argmax())], np.median(rel_ref), np.percentile(rel_ref, 90),
         rel_ref.max(), outliers, entries))
  assert np.median(rel) <= 2 * np.median(rel_ref) and np.percentile(rel, 90) <= 2 * np.percentile(rel_ref, 90) and rel.max() <= 3 * rel_ref.max(), \
      (tag, np.median(rel), rel.max())
  assert outliers <= 0.02 * entries, (tag, outliers, entries)


@pytest.mark.parametrize('how', ['graph', 'eager'])
def test_three_training_steps_follow_the_reference(golden, how):
  z = golden('model_steps_tiny.npz')
  net, left, right, gt, K = _setup(z)
  net.train()
  p0 = {k: v.detach().cpu().double().clone() for k, v in net.named_parameters()}
  count = data_parallel.global_valid_count(~torch.isnan(gt))
  losses = []
  if how == 'graph':  # bench.py's step
    reducer = data_parallel.GradAllReducer(net)
    opt = torch.optim.Adam(net.parameters(), lr=LR, betas=(0.9, 0.999), fused=True)
    # the capture's warm-up runs the body (BatchNorm state moves): snapshot and restore the module state around it
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}

    def body():
      reducer.zero_grad()
      loss, _ = net.forward_loss(left, right, gt, count=count)  # the loss and its gradient formed next to the heads, as bench.py runs it
      loss.backward()
      return loss

    graphed = GraphedStep(body, (left, right, gt, count), warmup=1)
    with torch.no_grad():
      for k, v in net.state_dict().items():
        v.copy_(state[k])
    for _ in range(K):
      loss = graphed.replay()
      reducer.all_reduce()
      opt.step()
      losses.append(float(loss))
  else:  # plain autograd accumulation, one launch per kernel, torch's ordinary Adam
    opt = torch.optim.Adam(net.parameters(), lr=LR, betas=(0.9, 0.999))
    for _ in range(K):
      net.train()
      opt.zero_grad()
      loss = _loss(net, left, right, gt, count)
      loss.backward()
      opt.step()
      losses.append(float(loss))
  torch.cuda.synchronize()
  _check(z, net, p0, losses, how)
