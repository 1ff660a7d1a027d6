"""GPU (-m gpu): every kernel of the 3-D stage at the shapes the benchmark actually runs.

bench.py's step (BASELINE configs[2]: Cassini 1024 x 512, 192 disparities) runs the 3-D kernels on 48 x 256 x 128, 24 x 128 x 64
and 12 x 64 x 32 volumes; the kernel tests in test_gpu_kernels.py use small ragged shapes.  Here each kernel runs at the benchmark
volume (one sample; the batch is the outermost loop of every kernel) and is compared with an fp64 CPU evaluation of the whole
layer -- oracle/conv_ref.py, 27 float64 GEMMs per layer, pinned against torch's conv3d in the CPU tier -- not with the vendor
library.  Tolerances are fp32 accumulation round-off of sums of Ci * 27 (forward / input gradient) or D*H*W (weight gradient)
terms, relative to the result's scale.
"""
import time

import numpy as np
import pytest
import torch

from oracle import conv_ref

import mode_hip
from mode_hip import functional as HF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

FULL = (48, 256, 128)
HALF = (24, 128, 64)
QUARTER = (12, 64, 32)


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()
  torch.set_num_threads(max(1, len(__import__('os').sched_getaffinity(0))))


def _rand(shape, seed, scale=1.0):
  return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


WGRAD = -1  # `terms` marker of a weight gradient


def _close(name, got, want, terms):
  """Forward / input gradient: |got - want| <= 2^-22 * sqrt(terms) * 2 * scale -- round-off of an fp32 sum of `terms` products
  (random-walk growth) relative to the largest magnitude of the exact result.  Weight gradient (terms = WGRAD): a sum
  over ~10^6 voxels whose exact value is itself ~sqrt(voxels) times a term, so the round-off relative to the result's scale does
  not grow with the volume: 1e-5 of the largest entry (recorded: 8.4e-7)."""
  got = got.detach().cpu().double()
  scale = max(1.0, float(want.abs().max()))
  err = float((got - want).abs().max())
  tol = (1e-5 if terms == WGRAD else 2.0**-22 * np.sqrt(terms) * 2) * scale  # (round 6: ~7-12 x the recorded errors; were 1e-4 / x 8)
  print('%s: max err %.3e (tol %.3e, scale %.3g)' % (name, err, tol, scale))
  assert err <= tol, (name, err, tol)


@pytest.mark.parametrize('Ci,Co,vol', [(32, 32, FULL), (64, 64, HALF), (64, 64, QUARTER)])
def test_conv3d_stride1_all_three_kernels(Ci, Co, vol, arith):
  """dres0/dres1/classifier-type 32 -> 32 layers at 48 x 256 x 128 and the hourglass' 64 -> 64 layers at 1/8 and 1/16."""
  D, H, W = vol
  x = _rand((1, Ci, D, H, W), 1)
  w = _rand((Co, Ci, 3, 3, 3), 2, (2.0 / (27 * Co))**0.5)
  gy = _rand((1, Co, D, H, W), 3)
  xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
  _close('conv3d_fwd %d->%d %s' % (Ci, Co, vol), HF.conv3d_fwd(xd, wd, 1), conv_ref.conv3d_fwd(x, w, 1), Ci * 27)
  _close('conv3d_bwd_data %d->%d %s' % (Ci, Co, vol), HF.conv3d_bwd_data(gd, wd, x.shape, 1), conv_ref.conv3d_bwd_data(gy, w, x.shape, 1),
         Co * 27)
  _close('conv3d_bwd_weight %d->%d %s' % (Ci, Co, vol), HF.conv3d_bwd_weight(gd, xd, 1), conv_ref.conv3d_bwd_weight(gy, x, 1), WGRAD)


@pytest.mark.parametrize('Ci,Co,vol', [(32, 64, FULL), (64, 64, HALF)])
def test_conv3d_stride2_all_three_kernels(Ci, Co, vol, arith):
  """hourglass conv1 (32 -> 64, 1/4 -> 1/8) and conv3 (64 -> 64, 1/8 -> 1/16)."""
  D, H, W = vol
  x = _rand((1, Ci, D, H, W), 4)
  w = _rand((Co, Ci, 3, 3, 3), 5, (2.0 / (27 * Co))**0.5)
  gy = _rand((1, Co, D // 2, H // 2, W // 2), 6)
  xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
  _close('conv3d_fwd s2 %d->%d' % (Ci, Co), HF.conv3d_fwd(xd, wd, 2), conv_ref.conv3d_fwd(x, w, 2), Ci * 27)
  _close('conv3d_bwd_data s2 %d->%d' % (Ci, Co), HF.conv3d_bwd_data(gd, wd, x.shape, 2), conv_ref.conv3d_bwd_data(gy, w, x.shape, 2), Co * 27)
  _close('conv3d_bwd_weight s2 %d->%d' % (Ci, Co), HF.conv3d_bwd_weight(gd, xd, 2), conv_ref.conv3d_bwd_weight(gy, x, 2), WGRAD)


@pytest.mark.parametrize('Cin,Cout,vol', [(64, 64, QUARTER), (64, 32, HALF)])
def test_deconv3d_forward_and_gradients(Cin, Cout, vol, arith):
  """hourglass conv5 (64 -> 64, 1/16 -> 1/8) and conv6 (64 -> 32, 1/8 -> 1/4): ConvTranspose3d k3 s2 p1 op1.  Its input gradient
  is the stride-2 convolution with the same weight, its weight gradient the stride-2 weight gradient with the roles exchanged
  (functional.Deconv3dFunction)."""
  D, H, W = vol
  x = _rand((1, Cin, D, H, W), 7)
  w = _rand((Cin, Cout, 3, 3, 3), 8, (2.0 / (27 * Cout))**0.5)
  gy = _rand((1, Cout, 2 * D, 2 * H, 2 * W), 9)
  xd = x.to(DEV).requires_grad_(True)
  wd = w.to(DEV).requires_grad_(True)
  y = HF.deconv3d(xd, wd)
  _close('deconv3d_fwd %d->%d' % (Cin, Cout), y, conv_ref.deconv3d_fwd(x, w), Cin * 27)
  y.backward(gy.to(DEV))
  _close('deconv3d input gradient', xd.grad, conv_ref.conv3d_fwd(gy, w, 2), Cout * 27)
  _close('deconv3d weight gradient', wd.grad, conv_ref.conv3d_bwd_weight(x, gy, 2), WGRAD)


def test_classifier_32_to_1_all_three_kernels():
  """classifN[2]: Conv3d(32 -> 1) at 48 x 256 x 128 (mode_disparity.py:76-80) -- the HBM-bound MFMA forms of csrc/conv3d_c1.hip."""
  D, H, W = FULL
  x = _rand((1, 32, D, H, W), 10)
  w = _rand((1, 32, 3, 3, 3), 11, (2.0 / 27)**0.5)
  gy = _rand((1, 1, D, H, W), 12)
  xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
  _close('conv3d_fwd 32->1', HF.conv3d_fwd(xd, wd, 1), conv_ref.conv3d_fwd(x, w, 1), 32 * 27)
  _close('conv3d_bwd_data 32->1', HF.conv3d_bwd_data(gd, wd, x.shape, 1), conv_ref.conv3d_bwd_data(gy, w, x.shape, 1), 27)
  _close('conv3d_bwd_weight 32->1', HF.conv3d_bwd_weight(gd, xd, 1), conv_ref.conv3d_bwd_weight(gy, x, 1), WGRAD)


@pytest.mark.parametrize('C,vol,relu,with_add', [(32, FULL, True, False), (32, FULL, False, True), (64, HALF, True, True)])
def test_batchnorm3d_train_forward_backward(C, vol, relu, with_add):
  """BatchNorm3d (+ residual add) (+ ReLU) in train mode at the benchmark volumes, batch 2, against torch's CPU BatchNorm in
  fp64: output, running statistics, and all gradients (csrc/bn_act.hip)."""
  import torch.nn as nn
  D, H, W = vol
  y = _rand((2, C, D, H, W), 13) * 1.7 + 0.3
  add = _rand((2, C, D, H, W), 14) if with_add else None
  go = _rand((2, C, D, H, W), 15)
  ref = nn.BatchNorm3d(C).double().train()
  with torch.no_grad():
    ref.weight.copy_(_rand((C,), 16) * 0.1 + 1)
    ref.bias.copy_(_rand((C,), 17) * 0.1)
  bn = nn.BatchNorm3d(C).to(DEV).train()
  with torch.no_grad():
    bn.weight.copy_(ref.weight.float())
    bn.bias.copy_(ref.bias.float())
  y64 = y.double().requires_grad_(True)
  a64 = add.double().requires_grad_(True) if with_add else None
  o64 = ref(y64)
  if with_add:
    o64 = o64 + a64
  if relu:
    o64 = torch.relu(o64)
  o64.backward(go.double())
  yd = y.to(DEV).requires_grad_(True)
  ad = add.to(DEV).requires_grad_(True) if with_add else None
  out = HF.bn_act(bn, yd, ad, relu)
  out.backward(go.to(DEV))
  n = 2 * D * H * W
  assert (out.detach().cpu().double() - o64.detach()).abs().max() < 2e-5
  assert (bn.running_mean.cpu().double() - ref.running_mean).abs().max() < 1e-6
  assert (bn.running_var.cpu().double() - ref.running_var).abs().max() < 1e-5
  assert int(bn.num_batches_tracked) == 1
  assert (yd.grad.cpu().double() - y64.grad).abs().max() < 2e-5
  if with_add:
    assert (ad.grad.cpu().double() - a64.grad).abs().max() < 1e-6
  for got, want in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
    assert (got.cpu().double() - want).abs().max() < 2.0**-22 * np.sqrt(n) * 8 * max(1.0, float(want.abs().max()))


def test_conv3d_stride2_weight_gradient_at_the_config4_volume():
  """32 -> 64 stride 2 at 64 x 512 x 256 (configs[4]'s quarter-resolution volume, one sample): 32 * D * H * W = 2^28 elements of x
  and 64 * 2^20 of gy still fit 32-bit lane offsets, so this layer must take the pipelined stride-2 kernel (it fell to the generic
  one while the limit was taken on max(Ci, Co) * D * H * W: 23 % instead of ~55 % of the MFMA peak) -- checked against the float64
  oracle."""
  D, H, W = 64, 512, 256
  x = _rand((1, 32, D, H, W), 11)
  gy = _rand((1, 64, D // 2, H // 2, W // 2), 12)
  got = HF.conv3d_bwd_weight(gy.to(DEV), x.to(DEV), 2)
  _close('conv3d_bwd_weight s2 32->64 at 64x512x256', got, conv_ref.conv3d_bwd_weight(gy, x, 2), WGRAD)


def test_whole_model_at_config4_per_gpu_share():
  """BASELINE configs[4]: 2048 x 1024 Cassini, 256 disparities, one pair per GPU -- the largest workload the north_star names
  (cost volume 2.15 GB if it were built; 64 x 512 x 256 quarter-resolution volume).  No CPU reference exists at this size
  (hours): the test holds the step to its size-independent properties -- finite, in range, deterministic run to run in the
  forward pass, a loss that falls when the step is applied, and a memory footprint that leaves the 288 GB untouched."""
  import models
  torch.manual_seed(0)
  torch.cuda.reset_peak_memory_stats()
  net = models.ModeDisparity(256, 'Sphere', 2048, 1024, 'Cassini').to(DEV).train()
  g = torch.Generator().manual_seed(1)
  left = torch.randn(1, 3, 2048, 1024, generator=g)
  right = torch.roll(left, -5, 3) + 0.01 * torch.randn(1, 3, 2048, 1024, generator=g)
  gt = torch.full((1, 1, 2048, 1024), 5.0)
  left, right, gt = left.to(DEV), right.to(DEV), gt.to(DEV)
  opt = torch.optim.SGD(net.parameters(), lr=1e-4)
  losses = []
  import no_vendor
  for _ in range(2):
    opt.zero_grad(set_to_none=True)
    with no_vendor.no_vendor_arithmetic():  # every layer of the largest configuration on the hand-written kernels
      preds = net(left, right)
    assert all(tuple(p.shape) == (1, 1, 2048, 1024) for p in preds)
    for p in preds:
      assert bool(torch.isfinite(p).all()) and float(p.min()) >= 0.0 and float(p.max()) <= 255.0
    loss = sum(wt * torch.nn.functional.smooth_l1_loss(p, gt) for wt, p in zip((0.5, 0.7, 1.0), preds))
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    opt.step()
    losses.append(float(loss))
  assert losses[1] < losses[0], losses
  net.eval()
  with torch.no_grad(), no_vendor.no_vendor_arithmetic():
    a = net(left, right)
    b = net(left, right)
  assert torch.equal(a, b)  # the eval forward is bit-reproducible
  peak = torch.cuda.max_memory_allocated() / 2**30
  print('configs[4] per-GPU share: losses %s, peak memory %.1f GB' % (losses, peak))
  assert peak < 100.0

