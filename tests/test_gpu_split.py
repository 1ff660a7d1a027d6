"""GPU (-m gpu): the stride-1 3x3x3 layers on the split-bf16 matrix path (csrc/conv3d_split.hip, functional.CONV_ARITH = 'bf16x6').

The claim under test is that this path computes an fp32 convolution: every check uses the SAME bound as the fp32 MFMA kernels'
tests (2^-22 * sqrt(terms) * 8 relative to the largest exact output, tests/test_gpu_fullsize.py), against float64 references
(torch's conv3d on the CPU for the small shapes, oracle/conv_ref.py at the benchmark size), and the error is printed next to the
fp32 kernel's on the same inputs.  Whole-model parity with the reference in this mode: tests/test_gpu_parity.py (parametrised
over both arithmetics)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv_ref

import mode_hip
from mode_hip import functional as HF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()


@pytest.fixture
def split_arith():
  HF.set_conv_arith('bf16x6')
  yield
  HF.set_conv_arith('bf16x6')


def _rand(shape, seed, scale=1.0):
  return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def _err(got, want):
  return float((got.detach().cpu().double() - want).abs().max())


def _tol(terms, want):
  return 2.0**-22 * np.sqrt(terms) * 8 * max(1.0, float(want.abs().max()))


CASES = [
    (2, 8, 8, 8, 8, 8),       # one chunk, 8 of 32 output rows, narrower than a column tile
    (1, 32, 32, 6, 10, 40),   # ragged W and H tiles
    (1, 64, 32, 4, 8, 32),    # dres0.0 (two chunk quartets)
    (2, 16, 20, 5, 7, 33),    # nothing divides anything
    (1, 32, 32, 3, 17, 130),  # odd depth, 5 column tiles
    (1, 32, 32, 12, 64, 64),  # several tiles per workgroup: the chunk stream crosses tile boundaries
    (3, 24, 32, 2, 8, 32),    # batch of three
]


@pytest.mark.parametrize('B,Ci,Co,D,H,W', CASES)
def test_split_forward_and_input_gradient_are_fp32_convolutions(B, Ci, Co, D, H, W, split_arith):
  assert mode_hip.lib().mode_conv3d_split_supported(Ci, Co, 1, 0) == 1
  x = _rand((B, Ci, D, H, W), 141)
  w = _rand((Co, Ci, 3, 3, 3), 142, (2.0 / (27 * Co))**0.5)
  gy = _rand((B, Co, D, H, W), 143)
  xa = x.double().requires_grad_(True)
  want = F.conv3d(xa, w.double(), None, 1, 1)
  want.backward(gy.double())
  xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
  y = HF.conv3d_fwd(xd, wd, 1)
  e_split, tol = _err(y, want.detach()), _tol(Ci * 27, want.detach())
  HF.set_conv_arith('f32')
  e_f32 = _err(HF.conv3d_fwd(xd, wd, 1), want.detach())
  HF.set_conv_arith('bf16x6')
  print('fwd %s: split %.3e, fp32 MFMA %.3e, bound %.3e' % ((B, Ci, Co, D, H, W), e_split, e_f32, tol))
  assert e_split <= tol
  assert torch.equal(y, HF.conv3d_fwd(xd, wd, 1)), 'not deterministic'
  if mode_hip.lib().mode_conv3d_split_supported(Ci, Co, 1, 1) == 1:
    gx = HF.conv3d_bwd_data(gd, wd, x.shape, 1)
    e_split, tol = _err(gx, xa.grad), _tol(Co * 27, xa.grad)
    print('bwd_data: split %.3e, bound %.3e' % (e_split, tol))
    assert e_split <= tol


WGRAD_CASES = [
    (2, 8, 8, 8, 8, 8),
    (1, 32, 32, 6, 10, 40),
    (1, 64, 32, 4, 8, 32),    # two x blocks
    (1, 32, 64, 4, 8, 32),    # two gy blocks
    (2, 20, 40, 5, 7, 33),    # partial blocks, odd width: the pair loads straddle the row end
    (1, 32, 32, 13, 64, 64),  # several units per workgroup, depth runs of different length
    (1, 16, 16, 1, 2, 31),    # a single depth
]


@pytest.mark.parametrize('B,Ci,Co,D,H,W', WGRAD_CASES)
def test_split_weight_gradient_is_an_fp32_weight_gradient(B, Ci, Co, D, H, W, split_arith):
  assert mode_hip.lib().mode_conv3d_split_supported(Ci, Co, 1, 2) == 1
  x = _rand((B, Ci, D, H, W), 171)
  gy = _rand((B, Co, D, H, W), 172)
  wa = torch.zeros((Co, Ci, 3, 3, 3), dtype=torch.float64, requires_grad=True)
  F.conv3d(x.double(), wa, None, 1, 1).backward(gy.double())
  want = wa.grad
  xd, gd = x.to(DEV), gy.to(DEV)
  got = HF.conv3d_bwd_weight(gd, xd, 1)
  HF.set_conv_arith('f32')
  got32 = HF.conv3d_bwd_weight(gd, xd, 1)
  HF.set_conv_arith('bf16x6')
  scale = max(1.0, float(want.abs().max()))
  e, e32 = _err(got, want), _err(got32, want)
  print('bwd_weight %s: split %.3e, fp32 MFMA %.3e (scale %.3g)' % ((B, Ci, Co, D, H, W), e, e32, scale))
  assert e <= 2e-5 * scale  # the bound of the fp32 kernels' test (tests/test_gpu_kernels.py::test_conv3d_fwd_bwd)
  assert torch.equal(got, HF.conv3d_bwd_weight(gd, xd, 1)), 'not deterministic'
  acc = torch.ones_like(got)
  HF.conv3d_bwd_weight(gd, xd, 1, into=acc)
  assert torch.allclose(acc, got + 1.0, rtol=0, atol=1e-5 * scale), 'accumulating form'


S2_WGRAD_CASES = [
    (2, 32, 64, 8, 12, 32),    # hourglass conv1 in small: x 32 channels, gy 64
    (1, 64, 64, 6, 8, 24),     # conv3; Wo = 12: a ragged 16-voxel column block
    (3, 64, 128, 2, 4, 8),     # two 64-channel gy blocks, one output depth
    (1, 32, 64, 14, 6, 48),    # 7 output depths in one unit: the odd tail of the two-phase loop
    (2, 32, 64, 40, 4, 16),    # long depth runs
]


@pytest.mark.parametrize('B,Ci,Co,D,H,W', S2_WGRAD_CASES)
def test_split_stride2_weight_gradient_against_float64(B, Ci, Co, D, H, W, split_arith):
  """mode_conv3d_bwd_weight_s2_split (csrc/conv3d_split_wgrad_s2.hip): the stride-2 layers' weight gradient on de-interleaved x rows
  and a ring of five planes, against float64 and beside the fp32 MFMA kernel on the same inputs; deterministic; accumulating form."""
  assert mode_hip.lib().mode_conv3d_split_supported(Ci, Co, 2, 2) == 1
  x = _rand((B, Ci, D, H, W), 271)
  gy = _rand((B, Co, D // 2, H // 2, W // 2), 272)
  wa = torch.zeros((Co, Ci, 3, 3, 3), dtype=torch.float64, requires_grad=True)
  F.conv3d(x.double(), wa, None, 2, 1).backward(gy.double())
  want = wa.grad
  xd, gd = x.to(DEV), gy.to(DEV)
  got = HF.conv3d_bwd_weight(gd, xd, 2)
  HF.set_conv_arith('f32')
  got32 = HF.conv3d_bwd_weight(gd, xd, 2)
  HF.set_conv_arith('bf16x6')
  scale = max(1.0, float(want.abs().max()))
  e, e32 = _err(got, want), _err(got32, want)
  print('stride-2 bwd_weight %s: split %.3e, fp32 MFMA %.3e (scale %.3g)' % ((B, Ci, Co, D, H, W), e, e32, scale))
  assert not torch.equal(got, got32), 'the split kernel ran (its rounding differs from the fp32 MFMA kernel\'s)'
  assert e <= 2e-5 * scale
  assert torch.equal(got, HF.conv3d_bwd_weight(gd, xd, 2)), 'not deterministic'
  acc = torch.ones_like(got)
  HF.conv3d_bwd_weight(gd, xd, 2, into=acc)
  assert torch.allclose(acc, got + 1.0, rtol=0, atol=1e-5 * scale), 'accumulating form'


def test_split_transposed_convolution_weight_gradient_against_float64(split_arith):
  """ConvTranspose3d k3 s2 p1 op1 (hourglass conv5 / conv6): its weight gradient is the same kernel with the operands exchanged."""
  for (cin, cout, D, H, W) in ((64, 32, 4, 6, 16), (64, 64, 3, 4, 8)):
    x = _rand((2, cin, D, H, W), 281)
    w = _rand((cin, cout, 3, 3, 3), 282, 0.05)
    xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want = F.conv_transpose3d(xa, wa, None, 2, 1, 1)
    gy = _rand(tuple(want.shape), 283)
    want.backward(gy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = HF.deconv3d(xd, wd)
    y.backward(gy.to(DEV))
    assert _err(y, want.detach()) <= _tol(cin * 27, want.detach())
    assert _err(wd.grad, wa.grad) <= 2e-5 * max(1.0, float(wa.grad.abs().max())), (cin, cout)
    assert _err(xd.grad, xa.grad) <= _tol(cout * 27, xa.grad)


def test_split_through_autograd_and_fallback_layers(split_arith):
  """HF.conv3d (the autograd op the model calls) in split mode: 64-channel layers run as two launches of 32 output channels; a
  stride-2 layer, a layer whose reduction channels are not a multiple of 8 and one with 96 output channels keep running on the fp32
  kernels."""
  lib = mode_hip.lib()
  assert lib.mode_conv3d_split_supported(32, 64, 1, 0) == 1 and lib.mode_conv3d_split_supported(64, 32, 1, 1) == 1
  assert lib.mode_conv3d_split_supported(32, 32, 2, 0) == 0 and lib.mode_conv3d_split_supported(32, 96, 1, 0) == 0
  assert lib.mode_conv3d_split_supported(12, 32, 1, 0) == 0  # reduction channels not a multiple of 8
  assert lib.mode_conv3d_split_supported(12, 32, 1, 2) == 1 and lib.mode_conv3d_split_supported(32, 1, 1, 2) == 0
  for (ci, co, stride) in ((64, 32, 1), (32, 64, 1), (64, 64, 1), (24, 40, 1), (32, 32, 2), (12, 32, 1)):
    x = _rand((1, ci, 4, 8, 32), 151)
    w = _rand((co, ci, 3, 3, 3), 152, 0.05)
    xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want = F.conv3d(xa, wa, None, stride, 1)
    gy = _rand(tuple(want.shape), 153)
    want.backward(gy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = HF.conv3d(xd, wd, stride)
    y.backward(gy.to(DEV))
    assert _err(y, want.detach()) <= _tol(ci * 27, want.detach()), (ci, co, stride)
    assert _err(xd.grad, xa.grad) <= _tol(co * 27, xa.grad), (ci, co, stride)
    assert _err(wd.grad, wa.grad) <= 2e-5 * max(1.0, float(wa.grad.abs().max())), (ci, co, stride)


@pytest.mark.parametrize('relu,with_add', [(True, False), (False, True), (True, True), (False, False)])
def test_split_with_the_folded_batchnorm_epilogue(relu, with_add, split_arith):
  with torch.no_grad():
    for ci, co in ((8, 32), (32, 20), (16, 64)):
      x, w = _rand((2, ci, 6, 10, 36), 161).to(DEV), _rand((co, ci, 3, 3, 3), 162, 0.1).to(DEV)
      bn = torch.nn.BatchNorm3d(co).to(DEV).eval()
      r = np.random.RandomState(163)
      bn.weight.copy_(torch.from_numpy(r.uniform(0.5, 1.5, co).astype(np.float32)))
      bn.bias.copy_(torch.from_numpy(r.standard_normal(co).astype(np.float32)))
      bn.running_mean.copy_(torch.from_numpy(r.standard_normal(co).astype(np.float32)))
      bn.running_var.copy_(torch.from_numpy(r.uniform(0.5, 2.0, co).astype(np.float32)))
      want = F.conv3d(x.cpu().double(), w.cpu().double(), None, 1, 1)
      want = (want - bn.running_mean.cpu().double().view(1, -1, 1, 1, 1)) / torch.sqrt(bn.running_var.cpu().double().view(1, -1, 1, 1, 1) + bn.eps)
      want = want * bn.weight.cpu().double().view(1, -1, 1, 1, 1) + bn.bias.cpu().double().view(1, -1, 1, 1, 1)
      add = _rand(tuple(want.shape), 164).to(DEV) if with_add else None
      if add is not None:
        want = want + add.cpu().double()
      if relu:
        want = torch.relu(want)
      got = HF.conv3d_bn_eval(x, w, bn, 1, add, relu)
      assert _err(got, want) < 1e-5 * max(1.0, float(want.abs().max())) * 4, (ci, co)


def test_split_at_the_benchmark_size_against_the_float64_oracle(split_arith):
  """32 -> 32 at 48 x 256 x 128 (one sample): forward and input gradient against oracle/conv_ref.py, same bound as the fp32 kernels'
  full-size test; the fp32 MFMA kernel's error on the same inputs is printed beside it."""
  D, H, W = 48, 256, 128
  torch.set_num_threads(max(1, len(__import__('os').sched_getaffinity(0))))
  x = _rand((1, 32, D, H, W), 1)
  w = _rand((32, 32, 3, 3, 3), 2, (2.0 / (27 * 32))**0.5)
  gy = _rand((1, 32, D, H, W), 3)
  xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
  want = conv_ref.conv3d_fwd(x, w, 1)
  got = HF.conv3d_fwd(xd, wd, 1)
  HF.set_conv_arith('f32')
  got32 = HF.conv3d_fwd(xd, wd, 1)
  HF.set_conv_arith('bf16x6')
  e, e32, tol = _err(got, want), _err(got32, want), _tol(32 * 27, want)
  rms = float((got.cpu().double() - want).pow(2).mean().sqrt())
  rms32 = float((got32.cpu().double() - want).pow(2).mean().sqrt())
  print('conv3d_fwd 32->32 full size: split max %.3e rms %.3e | fp32 MFMA max %.3e rms %.3e | bound %.3e' % (e, rms, e32, rms32, tol))
  assert e <= tol
  assert rms <= 1.25 * rms32, 'the split path must not be less accurate than the fp32 MFMA kernel'
  want = conv_ref.conv3d_bwd_data(gy, w, x.shape, 1)
  e = _err(HF.conv3d_bwd_data(gd, wd, x.shape, 1), want)
  print('conv3d_bwd_data 32->32 full size: split max %.3e (bound %.3e)' % (e, _tol(32 * 27, want)))
  assert e <= _tol(32 * 27, want)
  want = conv_ref.conv3d_bwd_weight(gy, x, 1)
  got = HF.conv3d_bwd_weight(gd, xd, 1)
  HF.set_conv_arith('f32')
  got32 = HF.conv3d_bwd_weight(gd, xd, 1)
  HF.set_conv_arith('bf16x6')
  scale = max(1.0, float(want.abs().max()))
  print('conv3d_bwd_weight 32->32 full size: split max %.3e, fp32 MFMA max %.3e (scale %.3g; bound 1e-4 * scale)' %
        (_err(got, want), _err(got32, want), scale))
  assert _err(got, want) <= 1e-4 * scale  # the bound of tests/test_gpu_fullsize.py for the fp32 kernels


# ------------------------------------------------------------------------------------------------ regular 3x3 Conv2d layers
CONV2D_CASES = [
    (2, 32, 32, 16, 64, 1),    # firstconv / layer1 shape, one output tile
    (1, 64, 64, 9, 40, 1),     # two output tiles in one launch, ragged tile
    (2, 16, 40, 7, 33, 1),     # partial second output tile, nothing divides anything
    (1, 64, 64, 12, 32, 2),    # dilation 2
    (1, 128, 128, 20, 32, 1),  # four output tiles = two launches
    (1, 32, 96, 18, 70, 2),    # three output tiles (one launch of two, one of one), dilation 2
    (4, 64, 64, 48, 64, 1),    # several tiles per workgroup: the chunk stream crosses tile boundaries
]


@pytest.mark.parametrize('B,Ci,Co,H,W,dil', CONV2D_CASES)
def test_split_conv2d_forward_and_input_gradient(B, Ci, Co, H, W, dil, split_arith):
  lib = mode_hip.lib()
  assert lib.mode_conv2d_split_supported(Ci, Co, dil, 0) == 1
  x = _rand((B, Ci, H, W), 181)
  w = _rand((Co, Ci, 3, 3), 182, (2.0 / (9 * Co))**0.5)
  gy = _rand((B, Co, H, W), 183)
  xa = x.double().requires_grad_(True)
  want = F.conv2d(xa, w.double(), None, 1, dil, dil)
  want.backward(gy.double())
  xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
  y = HF.conv2d_fwd(xd, wd, dil)
  HF.set_conv_arith('f32')
  y32 = HF.conv2d_fwd(xd, wd, dil)
  HF.set_conv_arith('bf16x6')
  e, e32, tol = _err(y, want.detach()), _err(y32, want.detach()), _tol(Ci * 9, want.detach())
  print('conv2d fwd %s: split %.3e, fp32 MFMA %.3e, bound %.3e' % ((B, Ci, Co, H, W, dil), e, e32, tol))
  assert e <= tol
  assert torch.equal(y, HF.conv2d_fwd(xd, wd, dil)), 'not deterministic'
  if lib.mode_conv2d_split_supported(Ci, Co, dil, 1) == 1:
    gx = HF.conv2d_bwd_data(gd, wd, dil)
    e, tol = _err(gx, xa.grad), _tol(Co * 9, xa.grad)
    print('conv2d bwd_data: split %.3e, bound %.3e' % (e, tol))
    assert e <= tol


@pytest.mark.parametrize('relu,with_add', [(True, False), (False, True), (True, True), (False, False)])
def test_split_conv2d_with_the_folded_batchnorm_epilogue(relu, with_add, split_arith):
  with torch.no_grad():
    for ci, co, dil in ((16, 32, 1), (32, 64, 2), (16, 128, 1)):
      x, w = _rand((2, ci, 20, 36), 191).to(DEV), _rand((co, ci, 3, 3), 192, 0.1).to(DEV)
      bn = torch.nn.BatchNorm2d(co).to(DEV).eval()
      r = np.random.RandomState(193)
      bn.weight.copy_(torch.from_numpy(r.uniform(0.5, 1.5, co).astype(np.float32)))
      bn.bias.copy_(torch.from_numpy(r.standard_normal(co).astype(np.float32)))
      bn.running_mean.copy_(torch.from_numpy(r.standard_normal(co).astype(np.float32)))
      bn.running_var.copy_(torch.from_numpy(r.uniform(0.5, 2.0, co).astype(np.float32)))
      want = F.conv2d(x.cpu().double(), w.cpu().double(), None, 1, dil, dil)
      v = lambda t: t.cpu().double().view(1, -1, 1, 1)
      want = (want - v(bn.running_mean)) / torch.sqrt(v(bn.running_var) + bn.eps) * v(bn.weight) + v(bn.bias)
      add = _rand(tuple(want.shape), 194).to(DEV) if with_add else None
      if add is not None:
        want = want + add.cpu().double()
      if relu:
        want = torch.relu(want)
      got = HF.conv2d_bn_eval(x, w, bn, dil, add, relu)
      assert _err(got, want) < 4e-5 * max(1.0, float(want.abs().max())), (ci, co, dil)


def test_split_conv2d_at_the_extractor_sizes_against_float64(split_arith):
  """64 -> 64 at 256 x 128 and 512 x 256, four images (the step's shapes): forward against torch's float64 conv2d on the CPU."""
  torch.set_num_threads(max(1, len(__import__('os').sched_getaffinity(0))))
  for H, W in ((256, 128), (512, 256)):
    x = _rand((4, 64, H, W), 201)
    w = _rand((64, 64, 3, 3), 202, (2.0 / (9 * 64))**0.5)
    want = F.conv2d(x.double(), w.double(), None, 1, 1)
    xd, wd = x.to(DEV), w.to(DEV)
    got = HF.conv2d_fwd(xd, wd, 1)
    HF.set_conv_arith('f32')
    got32 = HF.conv2d_fwd(xd, wd, 1)
    HF.set_conv_arith('bf16x6')
    rms = float((got.cpu().double() - want).pow(2).mean().sqrt())
    rms32 = float((got32.cpu().double() - want).pow(2).mean().sqrt())
    print('conv2d_fwd 64->64 %dx%d x 4: split max %.3e rms %.3e | fp32 MFMA max %.3e rms %.3e' % (H, W, _err(got, want), rms, _err(got32, want), rms32))
    assert _err(got, want) <= _tol(64 * 9, want)
    assert rms <= 1.25 * rms32


@pytest.mark.parametrize('B,Ci,Co,H,W,dil', CONV2D_CASES + [(2, 20, 40, 7, 33, 1), (1, 8, 8, 3, 5, 2), (3, 32, 32, 64, 96, 1)])
def test_split_conv2d_weight_gradient(B, Ci, Co, H, W, dil, split_arith):
  x = _rand((B, Ci, H, W), 211)
  gy = _rand((B, Co, H, W), 212)
  wa = torch.zeros((Co, Ci, 3, 3), dtype=torch.float64, requires_grad=True)
  F.conv2d(x.double(), wa, None, 1, dil, dil).backward(gy.double())
  want = wa.grad
  xd, gd = x.to(DEV), gy.to(DEV)
  got = HF.conv2d_bwd_weight(gd, xd, dil)
  HF.set_conv_arith('f32')
  got32 = HF.conv2d_bwd_weight(gd, xd, dil)
  HF.set_conv_arith('bf16x6')
  scale = max(1.0, float(want.abs().max()))
  e, e32 = _err(got, want), _err(got32, want)
  print('conv2d bwd_weight %s: split %.3e, fp32 MFMA %.3e (scale %.3g)' % ((B, Ci, Co, H, W, dil), e, e32, scale))
  assert e <= 2e-5 * scale
  assert torch.equal(got, HF.conv2d_bwd_weight(gd, xd, dil)), 'not deterministic'
  acc = torch.ones_like(got)
  HF.conv2d_bwd_weight(gd, xd, dil, into=acc)
  assert torch.allclose(acc, got + 1.0, rtol=0, atol=1e-5 * scale), 'accumulating form'


def test_split_kernels_on_random_small_shapes(split_arith):
  """Seeded random shapes around the tile edges (volumes smaller than a tile, widths 1..70, single rows / depths, partial channel
  blocks) for all five split kernel families, against torch's float64 convolutions."""
  r = np.random.RandomState(2022)
  for it in range(24):
    B = int(r.randint(1, 3))
    Ci3, Co3 = 8 * int(r.randint(1, 5)), int(r.choice([8, 20, 32, 40, 64]))
    D, H, W = int(r.randint(1, 6)), int(r.randint(1, 20)), int(r.randint(1, 71))
    x = _rand((B, Ci3, D, H, W), 300 + it)
    w = _rand((Co3, Ci3, 3, 3, 3), 400 + it, 0.1)
    gy = _rand((B, Co3, D, H, W), 500 + it)
    xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want = F.conv3d(xa, wa, None, 1, 1)
    want.backward(gy.double())
    xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
    tag = ('3d', B, Ci3, Co3, D, H, W)
    assert _err(HF.conv3d_fwd(xd, wd, 1), want.detach()) <= _tol(Ci3 * 27, want.detach()), tag
    if Co3 % 8 == 0:
      assert _err(HF.conv3d_bwd_data(gd, wd, x.shape, 1), xa.grad) <= _tol(Co3 * 27, xa.grad), tag
    assert _err(HF.conv3d_bwd_weight(gd, xd, 1), wa.grad) <= 2e-5 * max(1.0, float(wa.grad.abs().max())), tag
    Ci2, Co2, dil = 16 * int(r.randint(1, 5)), int(r.choice([16, 24, 32, 64, 96])), int(r.randint(1, 3))
    H2, W2 = int(r.randint(1, 40)), int(r.randint(1, 71))
    x = _rand((B, Ci2, H2, W2), 600 + it)
    w = _rand((Co2, Ci2, 3, 3), 700 + it, 0.1)
    gy = _rand((B, Co2, H2, W2), 800 + it)
    xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want = F.conv2d(xa, wa, None, 1, dil, dil)
    want.backward(gy.double())
    xd, wd, gd = x.to(DEV), w.to(DEV), gy.to(DEV)
    tag = ('2d', B, Ci2, Co2, H2, W2, dil)
    assert _err(HF.conv2d_fwd(xd, wd, dil), want.detach()) <= _tol(Ci2 * 9, want.detach()), tag
    if Co2 % 16 == 0:
      assert _err(HF.conv2d_bwd_data(gd, wd, dil), xa.grad) <= _tol(Co2 * 9, xa.grad), tag
    assert _err(HF.conv2d_bwd_weight(gd, xd, dil), wa.grad) <= 2e-5 * max(1.0, float(wa.grad.abs().max())), tag


# ------------------------------------------------------------------------------------------------ spherical input gradient (a8)
@pytest.mark.parametrize('ih,iw,B,ci,co,groups', [(128, 256, 2, 128, 128, 1), (128, 256, 4, 64, 128, 1), (128, 256, 1, 48, 32, 2), (32, 64, 3, 40, 16, 1)])
def test_split_sphere_input_gradient_against_float64(ih, iw, B, ci, co, groups, split_arith, monkeypatch):
  """mode_sphere_conv_bwd_data_win_split (windowed adjoint, split-bf16) + the gather kernel on the left-over tile list against the
  float64 oracle (oracle/sphere_conv_ref.py: cu:293-356 + cpp:275-315 restated), at the benchmark's quarter-resolution Cassini grid
  256 x 128 (128 -> 128 and the 64 -> 128 layer, groups, channel counts off the 128 / 32 blocks) and at 64 x 32 (5 good tiles);
  same bound as the fp32 gather kernel, whose error on the same inputs is printed beside it; written (not added to), deterministic."""
  from oracle import mode_ref, sphere_conv_ref
  monkeypatch.setattr(HF, 'SPHERE_BWD_SPLIT_MIN_WG', 0)
  pos = mode_ref.sphere_position(ih, iw, 'Cassini').contiguous()
  H, W = pos.shape[2:]
  w = _rand((co, ci // groups, 3, 3), 401, (2.0 / (9 * ci // groups))**0.5)
  gy = _rand((B, co, H, W), 402)
  x0 = torch.zeros((B, ci, H, W), dtype=torch.float64)
  torch.set_num_threads(max(1, len(__import__('os').sched_getaffinity(0))))
  want, _ = sphere_conv_ref.backward(x0, pos, w.double(), gy.double(), (1, 1), (1, 1), (1, 1), groups)
  pd, wd = pos.to(DEV), w.to(DEV)
  ap = HF.sphere_adjplan(pd, 3, 3)
  assert ap is not None and ap[1] > 0, 'the windowed adjoint must be planned for this table'
  assert mode_hip.lib().mode_sphere_conv_bwd_data_win_supported(ci, co, groups) == 1
  gyt = gy.to(DEV).transpose(2, 3).contiguous()
  gxt = torch.full((B, ci, W, H), float('nan'), device=DEV)
  HF.sphere_conv_bwd_data_t(gyt, pd, wd, gxt, groups)
  got = gxt.transpose(2, 3)
  monkeypatch.setattr(HF, 'SPHERE_BWD_DATA_SPLIT', False)
  old = torch.full((B, ci, W, H), float('nan'), device=DEV)
  HF.sphere_conv_bwd_data_t(gyt, pd, wd, old, groups)
  monkeypatch.setattr(HF, 'SPHERE_BWD_DATA_SPLIT', True)
  e, e_old = _err(got, want), _err(old.transpose(2, 3), want)
  tol = 2e-6 * (co // groups * 9) * max(1.0, float(want.abs().max()))
  rms = float((got.cpu().double() - want).pow(2).mean().sqrt())
  rms_old = float((old.transpose(2, 3).cpu().double() - want).pow(2).mean().sqrt())
  print('sphere_conv_bwd_data %d->%d %dx%d B=%d g=%d: split max %.3e rms %.3e | fp32 gather kernel max %.3e rms %.3e | bound %.3e; %d of %d tiles on '
        'the split kernel' % (ci, co, H, W, B, groups, e, rms, e_old, rms_old, tol, ap[1], ap[1] + ap[5] // 4))
  assert torch.isfinite(got).all()
  assert e <= tol
  assert rms <= 1.25 * rms_old + 1e-12, 'the split path must not be less accurate than the fp32 kernel'
  again = torch.empty_like(gxt)
  HF.sphere_conv_bwd_data_t(gyt, pd, wd, again, groups)
  assert torch.equal(again, gxt)  # deterministic
  # the NCHW operator (reference seam) takes the same route through its transposed copies
  gx = torch.full((B, ci, H, W), float('nan'), device=DEV)
  HF.sphere_conv_bwd_data(gy.to(DEV), pd, wd, gx, (1, 1), groups, overwrite=True, gy_transposed=gyt)
  assert torch.equal(gx, got)


@pytest.mark.parametrize('ih,iw,B,ci,co,groups', [(128, 256, 2, 128, 128, 1), (128, 256, 4, 64, 128, 1), (128, 256, 1, 40, 24, 1), (128, 256, 1, 48, 200, 1),
                                                   (64, 128, 2, 64, 64, 2)])
def test_split_sphere_weight_gradient_against_float64(ih, iw, B, ci, co, groups, split_arith, monkeypatch):
  """sphere_bww_split_kernel (K = 16 pixels per bf16 MFMA, split operands) + the polar kernel + the reduction against the float64
  oracle, on plane-transposed storage as the model uses it: benchmark shape 128 -> 128 and 64 -> 128 at 256 x 128, channel counts off
  the 32 / 128 blocks (masked), more than 128 output channels (two slices), groups; same bound as the fp32 windowed kernel, whose
  error on the same inputs is printed beside it; adds to gw (reference contract), deterministic."""
  from oracle import mode_ref, sphere_conv_ref
  pos = mode_ref.sphere_position(ih, iw, 'Cassini').contiguous()
  H, W = pos.shape[2:]
  x = _rand((B, ci, H, W), 411)
  gy = _rand((B, co, H, W), 412)
  w0 = torch.zeros((co, ci // groups, 3, 3), dtype=torch.float64)
  torch.set_num_threads(max(1, len(__import__('os').sched_getaffinity(0))))
  _, want = sphere_conv_ref.backward(x.double(), pos, w0, gy.double(), (1, 1), (1, 1), (1, 1), groups)
  pd = pos.to(DEV)
  plan = HF.sphere_plan(pd, 3, 3)
  assert plan is not None and plan[1][0] > 0
  xt, gyt = x.to(DEV).transpose(2, 3).contiguous(), gy.to(DEV).transpose(2, 3).contiguous()
  gw = torch.zeros((co, ci // groups, 3, 3), device=DEV)
  HF.sphere_conv_bwd_weight_t(gyt, pd, xt, gw, groups)
  monkeypatch.setattr(HF, 'SPHERE_BWD_WEIGHT_SPLIT', False)
  old = torch.zeros_like(gw)
  HF.sphere_conv_bwd_weight_t(gyt, pd, xt, old, groups)
  monkeypatch.setattr(HF, 'SPHERE_BWD_WEIGHT_SPLIT', True)
  scale = max(1.0, float(want.abs().max()))
  e, e_old = _err(gw, want), _err(old, want)
  rms = float((gw.cpu().double() - want).pow(2).mean().sqrt())
  rms_old = float((old.cpu().double() - want).pow(2).mean().sqrt())
  print('sphere_conv_bwd_weight %d->%d %dx%d B=%d g=%d: split max %.3e rms %.3e | fp32 windowed kernel max %.3e rms %.3e | bound %.3e (|gw| <= %.3g)' %
        (ci, co, H, W, B, groups, e, rms, e_old, rms_old, 1e-5 * scale, scale))
  assert e <= 1e-5 * scale
  assert rms <= 1.5 * rms_old + 1e-12
  HF.sphere_conv_bwd_weight_t(gyt, pd, xt, gw, groups)  # a second call adds on top (sphere_conv.py:62-64)
  assert _err(gw, 2 * want) <= 2e-5 * scale
  again = torch.zeros_like(old)
  HF.sphere_conv_bwd_weight_t(gyt, pd, xt, again, groups)
  again2 = torch.zeros_like(old)
  HF.sphere_conv_bwd_weight_t(gyt, pd, xt, again2, groups)
  assert torch.equal(again, again2)  # deterministic
