"""GPU (-m gpu): the fused classifier head of the training step (csrc/classif_head.hip, functional.ClassifHeadFunction) --
BatchNorm3d (batch statistics) + ReLU + Conv3d(C -> 1) [+ residual] from the first convolution's output on
(models/mode_disparity.py:76-80, 127-129) -- against a float64 composition of torch's own CPU operators (output, running statistics,
every gradient), at ragged shapes, at the benchmark volume, and against the library's own unfused composition."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import mode_hip
from mode_hip import functional as HF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()


def _rand(shape, seed, scale=1.0):
  return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def _modules(C, seed):
  bn = nn.BatchNorm3d(C).train()
  conv = nn.Conv3d(C, 1, 3, padding=1, bias=False)
  with torch.no_grad():
    bn.weight.copy_(_rand((C,), seed) * 0.2 + 1)
    bn.bias.copy_(_rand((C,), seed + 1) * 0.2)
    conv.weight.copy_(_rand((1, C, 3, 3, 3), seed + 2, (2.0 / 27)**0.5))
  return bn, conv


def _reference(y, add, go, bn, conv):
  """float64 on the CPU: returns (cost, gy, gadd, gw, ggamma, gbeta, running_mean, running_var)."""
  b64, c64 = nn.BatchNorm3d(bn.num_features).double().train(), nn.Conv3d(bn.num_features, 1, 3, padding=1, bias=False).double()
  with torch.no_grad():
    b64.weight.copy_(bn.weight.double())
    b64.bias.copy_(bn.bias.double())
    c64.weight.copy_(conv.weight.double())
  y64 = y.double().requires_grad_(True)
  a64 = add.double().requires_grad_(True) if add is not None else None
  pre = b64(y64)
  cost = c64(torch.relu(pre))
  if a64 is not None:
    cost = cost + a64
  cost.backward(go.double())
  return (cost.detach(), y64.grad, a64.grad if a64 is not None else None, c64.weight.grad, b64.weight.grad, b64.bias.grad, b64.running_mean,
          b64.running_var, pre.detach())


def _check_off_the_relu_threshold(name, got, want, pre, tol):
  """dL/dy of an element carries its ReLU mask: where the float64 pre-activation is within fp32 round-off of zero the two evaluations may
  disagree about the mask (a handful of 10^8 elements at the benchmark volume) -- those elements are left out, and counted."""
  near = pre.abs() <= 1e-5
  n_near = int(near.sum())
  print('%s: %d of %d elements within 1e-5 of the ReLU threshold left out' % (name, n_near, near.numel()))
  assert n_near <= 1e-4 * near.numel()
  diff = (got.detach().cpu().double() - want).abs()
  diff[near] = 0
  err, scale = float(diff.max()), max(1.0, float(want.abs().max()))
  print('%s: max err %.3e (tol %.3e x scale %.3g)' % (name, err, tol, scale))
  assert err <= tol * scale, (name, err, tol, scale)


def _fused(y, add, go, bn, conv):
  bn, conv = bn.to(DEV), conv.to(DEV)
  yd = y.to(DEV).requires_grad_(True)
  ad = add.to(DEV).requires_grad_(True) if add is not None else None
  assert HF.classif_fused_supported(yd, bn, conv)
  cost = HF.classif_head_train(yd, bn, conv, ad)
  cost.backward(go.to(DEV))
  torch.cuda.synchronize()
  return cost.detach(), yd.grad, ad.grad if ad is not None else None, conv.weight.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var


def _check(name, got, want, tol):
  err = float((got.detach().cpu().double() - want).abs().max())
  scale = max(1.0, float(want.abs().max()))
  print('%s: max err %.3e (tol %.3e x scale %.3g)' % (name, err, tol, scale))
  assert err <= tol * scale, (name, err, tol, scale)


@pytest.mark.parametrize('B,C,vol,with_add', [(2, 32, (4, 16, 32), False), (2, 5, (5, 11, 37), True), (1, 32, (13, 20, 70), True),
                                               (3, 17, (2, 9, 33), False)])
def test_fused_classifier_head_small_and_ragged(B, C, vol, with_add):
  D, H, W = vol
  y = _rand((B, C, D, H, W), 1) * 1.7 + 0.4
  add = _rand((B, 1, D, H, W), 2) if with_add else None
  go = _rand((B, 1, D, H, W), 3)
  bn, conv = _modules(C, 10)
  want = _reference(y, add, go, bn, conv)
  got = _fused(y, add, go, bn, conv)
  n = B * D * H * W
  _check('cost', got[0], want[0], 2.0**-22 * np.sqrt(27 * C) * 8)
  _check_off_the_relu_threshold('gy', got[1], want[1], want[8], 2e-5)
  if with_add:
    _check('gadd', got[2], want[2], 1e-6)
  _check('gw', got[3], want[3], 2.0**-22 * np.sqrt(n) * 8)
  _check('ggamma', got[4], want[4], 2.0**-22 * np.sqrt(n * 27) * 8)
  _check('gbeta', got[5], want[5], 2.0**-22 * np.sqrt(n * 27) * 8)
  _check('running_mean', got[6], want[6], 1e-6)
  _check('running_var', got[7], want[7], 1e-5)
  assert int(bn.num_batches_tracked) == 1


def test_fused_classifier_head_with_a_large_mean():
  """|mean| >> std in the BatchNorm input: the sums of the backward are taken about the batch mean, so nothing cancels."""
  B, C, D, H, W = 2, 8, 4, 16, 32
  y = _rand((B, C, D, H, W), 4) * 0.05 + 30.0
  go = _rand((B, 1, D, H, W), 5)
  bn, conv = _modules(C, 20)
  want = _reference(y, None, go, bn, conv)
  got = _fused(y, None, go, bn, conv)
  _check('cost', got[0], want[0], 2e-3)  # (the normalised values themselves carry 30 / 0.05 * 2^-24 = 4e-5 each)
  _check('gw', got[3], want[3], 2e-3)
  _check('ggamma', got[4], want[4], 2e-3)
  _check('gbeta', got[5], want[5], 1e-4)


def test_fused_classifier_head_at_the_benchmark_volume():
  """2 x 32 x 48 x 256 x 128 (BASELINE configs[2]) against float64 and against the library's unfused operators on the same inputs."""
  B, C, D, H, W = 2, 32, 48, 256, 128
  torch.set_num_threads(max(1, len(__import__('os').sched_getaffinity(0))))
  y = _rand((B, C, D, H, W), 6) * 1.3 + 0.2
  add = _rand((B, 1, D, H, W), 7)
  go = _rand((B, 1, D, H, W), 8)
  bn, conv = _modules(C, 30)
  want = _reference(y, add, go, bn, conv)
  import copy
  bn2, conv2 = copy.deepcopy(bn).to(DEV), copy.deepcopy(conv).to(DEV)
  got = _fused(y, add, go, bn, conv)
  n = B * D * H * W
  _check('cost', got[0], want[0], 2.0**-22 * np.sqrt(27 * C) * 8)
  _check_off_the_relu_threshold('gy', got[1], want[1], want[8], 2e-5)
  _check('gadd', got[2], want[2], 1e-6)
  _check('gw', got[3], want[3], 1e-4)
  # (the two BatchNorm gradients carry the ReLU masks of 10^8 elements: a mask that differs between the float32 and the float64 evaluation
  # of an element within round-off of the threshold moves them by |g| ~ 1 -- same bound as test_batchnorm3d_train_forward_backward)
  _check('ggamma', got[4], want[4], 2.0**-22 * np.sqrt(n) * 8)
  _check('gbeta', got[5], want[5], 2.0**-22 * np.sqrt(n) * 8)
  _check('running_mean', got[6], want[6], 1e-6)
  _check('running_var', got[7], want[7], 1e-5)
  # the composition of separate operators the fused head replaces
  yd = y.to(DEV).requires_grad_(True)
  ad = add.to(DEV).requires_grad_(True)
  cost = HF.conv3d(HF.bn_act(bn2, yd, None, True), conv2.weight, 1) + ad
  cost.backward(go.to(DEV))
  torch.cuda.synchronize()
  _check('cost vs unfused', got[0], cost.detach().cpu().double(), 1e-5)
  _check('gy vs unfused', got[1], yd.grad.cpu().double(), 2e-5)  # (the same float32 mask expression with the same coefficients: no flips)
  _check('gw vs unfused', got[3], conv2.weight.grad.cpu().double(), 1e-4)
  _check('ggamma vs unfused', got[4], bn2.weight.grad.cpu().double(), 1e-4)
  _check('gbeta vs unfused', got[5], bn2.bias.grad.cpu().double(), 1e-4)


def test_model_uses_the_fused_head_and_matches_the_unfused_composition():
  """ModeDisparity in training mode at 64 x 32 / 16: the three heads run ClassifHeadFunction, and predictions and gradients equal those
  of the same model with HF.CLASSIF_FUSED switched off to fp32 round-off."""
  import sys, os
  sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
  import recipe
  import models
  state = recipe.recipe_state_wc(recipe.load_manifest(), 100)
  left, right = recipe.recipe_images(2, 64, 32, 101)
  left, right = left.to(DEV), right.to(DEV)
  res = {}
  for fused in (True, False):
    HF.CLASSIF_FUSED = fused
    try:
      net = models.ModeDisparity(16, 'Sphere', 64, 32, 'Cassini').to(DEV)
      net.load_state_dict(state)
      net.train()
      preds = net(left, right)
      names = set()
      fn_stack = [p.grad_fn for p in preds]
      seen = set()
      while fn_stack:
        f = fn_stack.pop()
        if f is None or f in seen:
          continue
        seen.add(f)
        names.add(type(f).__name__)
        fn_stack.extend(g for g, _ in f.next_functions)
      assert ('ClassifHeadFunctionBackward' in names) == fused
      sum(float(wt) * p.abs().mean() for wt, p in zip((0.5, 0.7, 1.0), preds)).backward()
      res[fused] = ([p.detach().clone() for p in preds], {k: v.grad.detach().clone() for k, v in net.named_parameters()},
                    {k: v.detach().clone() for k, v in net.named_buffers() if 'classif' in k})
    finally:
      HF.CLASSIF_FUSED = True
  for a, b in zip(res[True][0], res[False][0]):
    assert float((a - b).abs().max()) <= 2e-5
  num = sum(float(((res[True][1][k] - res[False][1][k]).double()**2).sum()) for k in res[True][1])
  den = sum(float((res[False][1][k].double()**2).sum()) for k in res[True][1])
  print('whole-network gradient, fused vs unfused heads: relative L2 %.3e' % (num / den)**0.5)
  assert (num / den)**0.5 < 1e-4
  for k, v in res[True][2].items():
    assert float((v.double() - res[False][2][k].double()).abs().max()) <= 1e-6 * max(1.0, float(v.double().abs().max())), k
