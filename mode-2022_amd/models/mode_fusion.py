"""Fusion stage of MODE, MI355X edition: drop-in for the reference's ``models/mode_fusion.py``.

Same public classes, constructor arguments, module tree (hence state_dict keys and shapes) and initialisation as the
reference: ``ModeFusion(maxdepth, channels, inplanes)`` (mode_fusion.py:286-307, built as
``ModeFusion(args.maxdepth, [32, 64, 128, 256], {'depth': 12, 'rgb': 12})`` in train_fusion.py:64), ``Baseline(maxdepth)``
(:269-283) and ``depth_regression`` (:255-265).

The network is a 2D U-Net of 3x3 ``Conv2d -> BatchNorm2d -> ReLU`` pairs with max-pooling on the way down and 2x2 transposed
convolutions on the way up.  Since round 6 every layer runs on the hand-written kernels (no MIOpen / rocBLAS call in its forward or
backward; ``tests/test_fusion.py`` runs it under the no-vendor guard): the 3x3 convolutions on the disparity stage's ``conv2d``
kernels (split-bf16 matrix path; the 12-channel input layers on the fp32 MFMA kernel; inference folds the BatchNorm into the
convolution), every ``BatchNorm2d (+ ReLU)`` on the fused BatchNorm kernels, max-pooling / the rearrangement half of the 2x2
transposed convolutions / the sigmoid head on ``csrc/fusion_ops.hip``, the GEMM half of the transposed convolutions on
``csrc/conv1x1.hip``.  What is left to ATen is data movement (``cat``) and the final multiplication by ``maxdepth``.
CPU tensors raise NotImplementedError like the rest of the package: there is no CPU path."""
import math

import torch
import torch.nn as nn

from mode_hip import functional as HF

from . import stage3d


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
  """Conv2d (no bias) + BatchNorm2d, mode_fusion.py:12-14."""
  return nn.Sequential(
      nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=dilation if dilation > 1 else pad,
                dilation=dilation, bias=False), nn.BatchNorm2d(out_planes))


def _run(module, x):
  """Evaluate a (nested) Sequential on the hand-written kernels: convbn (+ ReLU) as stage3d.conv_bn (training: convolution kernel +
  the fused BatchNorm passes; inference: ONE launch, BatchNorm folded into the convolution), max-pooling, the 2x2 transposed
  convolution with its BatchNorm + ReLU, and the single-channel head with its sigmoid (csrc/fusion_ops.hip).  Host tensors keep the
  torch modules (the CPU wiring tests)."""
  if isinstance(module, BasicBlock):
    return module(x)
  if not isinstance(module, nn.Sequential):
    return module(x)
  items = list(module.children())
  i = 0
  while i < len(items):
    m = items[i]
    nxt = items[i + 1] if i + 1 < len(items) else None
    if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.BatchNorm2d):
      relu = isinstance(nxt, nn.ReLU)  # convbn (+ ReLU)
      x = stage3d.conv_bn(m, x, relu=relu) if x.is_cuda else stage3d.bn_act(m[1], m[0](x), None, relu)
      i += 2 if relu else 1
    elif isinstance(m, nn.ConvTranspose2d) and isinstance(nxt, nn.BatchNorm2d) and x.is_cuda and HF.deconv2x2_supported(x, m):
      relu = i + 2 < len(items) and isinstance(items[i + 2], nn.ReLU)  # ConvTranspose2d -> BatchNorm2d -> ReLU (mode_fusion.py:195-197)
      if HF.bn_foldable(nxt):
        x = HF.deconv2x2_bn_eval(x, m, nxt, relu)
      else:
        x = stage3d.bn_act(nxt, HF.deconv2x2(x, m), None, relu)
      i += 3 if relu else 2
    elif isinstance(m, nn.BatchNorm2d):
      relu = isinstance(nxt, nn.ReLU)
      x = stage3d.bn_act(m, x, None, relu)
      i += 2 if relu else 1
    elif isinstance(m, nn.MaxPool2d) and x.is_cuda and HF.maxpool2x2_supported(x, m):
      x = HF.maxpool2x2(x)
      i += 1
    elif isinstance(m, nn.Conv2d) and isinstance(nxt, nn.Sigmoid) and x.is_cuda and HF.conv1x1_sigmoid_supported(x, m):
      x = HF.conv1x1_sigmoid(x, m)  # Conv2d(planes, 1, 1, bias=True) + Sigmoid (mode_fusion.py:228-229)
      i += 2
    else:
      x = _run(m, x)
      i += 1
  return x


class BasicBlock(nn.Module):
  """Two convbn + ReLU pairs; `downsample` is accepted and unused, as in the reference (mode_fusion.py:17-34) -- the 1x1
  projections the reference builds for it are never registered (same state_dict; see _rng_parity_downsample)."""
  expansion = 1

  def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
    super(BasicBlock, self).__init__()
    self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
    self.conv2 = nn.Sequential(convbn(planes, planes, 3, 1, pad, dilation), nn.ReLU(inplace=True))
    self.stride = stride

  def forward(self, x):
    return _run(self.conv2, _run(self.conv1, x))


def _rng_parity_downsample(inplanes, planes, stride):
  """The reference builds a 1x1 `downsample` projection for the first block of a layer (mode_fusion.py:50-54, 117-120, ...) that
  BasicBlock never registers or uses.  It is built and dropped here too, for one reason only: its default initialisation draws from
  the global RNG, and without those draws the ConvTranspose2d / bias defaults and the He-normal re-initialisation that follow would
  differ from the reference's under the same torch.manual_seed."""
  nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)


def _blocks(inplanes, planes, blocks, stride, pad, dilation):
  if stride != 1 or inplanes != planes:
    _rng_parity_downsample(inplanes, planes, stride)
  layers = [BasicBlock(inplanes, planes, stride, None, pad, dilation)]
  layers += [BasicBlock(planes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
  return layers


def _head(planes):
  return [nn.Conv2d(planes, 1, kernel_size=1, padding=0, stride=1, bias=True), nn.Sigmoid()]


def _up(planes):
  return [nn.ConvTranspose2d(planes, int(planes / 2), 2, 2), nn.BatchNorm2d(int(planes / 2)), nn.ReLU(inplace=True)]


class feature_extraction_Baseline(nn.Module):
  """mode_fusion.py:37-92: a plain stack on the 6 concatenated depth maps."""

  def __init__(self, maxdepth):
    super(feature_extraction_Baseline, self).__init__()
    self.inplanes = 6
    widths = (32, 64, 128, 256, 128, 64)
    for i, (planes, blocks) in enumerate(zip(widths, (2, 1, 1, 1, 1, 1)), 1):
      setattr(self, 'layer%d' % i, nn.Sequential(*_blocks(self.inplanes, planes, blocks, 1, 1, 1)))
      self.inplanes = planes
    self.layer7 = nn.Sequential(*(_blocks(self.inplanes, 32, 2, 1, 1, 1) + _head(32)))
    self.inplanes = 32
    self.maxdepth = torch.tensor(maxdepth)

  def forward(self, x):
    for i in range(1, 8):
      x = _run(getattr(self, 'layer%d' % i), x)
    return x * self.maxdepth


class feature_extraction_MODE_Fusion(nn.Module):
  """mode_fusion.py:95-252: depth(+confidence) encoder, RGB encoder, per-scale fusion blocks, decoder with skip connections."""

  def __init__(self, maxdepth, channels, inplanes):
    super(feature_extraction_MODE_Fusion, self).__init__()
    c = channels
    self.depth_inplanes = inplanes['depth']
    self.rgb_inplanes = inplanes['rgb']
    pool = lambda: [nn.MaxPool2d(2, stride=2)]  # noqa: E731

    def take(kind, planes, blocks, before=(), after=()):
      cin = getattr(self, kind + '_inplanes')
      body = _blocks(cin, planes, blocks, 1, 1, 1)
      seq = nn.Sequential(*(list(before) + body + (after() if callable(after) else list(after))))  # construction (= RNG) order of the reference
      setattr(self, kind + '_inplanes', planes)
      return seq

    # registration order = the reference's (it fixes the order of the state_dict)
    self.depth_layer1 = take('depth', c[0], 2)
    self.depth_layer2 = take('depth', c[1], 1, before=pool())
    self.depth_layer3 = take('depth', c[2], 1, before=pool())
    self.rgb_layer1 = take('rgb', c[0], 2)
    self.rgb_layer2 = take('rgb', c[1], 1, before=pool())
    self.rgb_layer3 = take('rgb', c[2], 1, before=pool())
    self.fusion_layer1 = self._make_fusion_layer(c[0], 2)
    self.fusion_layer2 = self._make_fusion_layer(c[1], 2)
    self.fusion_layer3 = self._make_fusion_layer(c[2], 2)
    self.depth_layer4 = take('depth', c[3], 1, before=pool(), after=lambda: _up(c[3]))
    self.depth_layer5 = take('depth', c[2], 1, after=lambda: _up(c[2]))
    self.depth_layer6 = take('depth', c[1], 1, after=lambda: _up(c[1]))
    self.depth_layer7 = take('depth', c[0], 2, after=lambda: _head(c[0]))
    self.maxdepth = torch.tensor(maxdepth)

  @staticmethod
  def _make_fusion_layer(planes, blocks):
    """mode_fusion.py:176-184: blocks on the concatenation of the depth and RGB features of one scale."""
    _rng_parity_downsample(int(2 * planes), planes, 1)  # (unconditional in the reference)
    return nn.Sequential(*([BasicBlock(int(2 * planes), planes, 1, None, 1, 1)] + [BasicBlock(planes, planes, 1, None, 1, 1)
                                                                                      for _ in range(1, blocks)]))

  def forward(self, depth_input, rgb_input):
    depth1 = _run(self.depth_layer1, depth_input)
    depth2 = _run(self.depth_layer2, depth1)
    depth3 = _run(self.depth_layer3, depth2)
    depth4 = _run(self.depth_layer4, depth3)
    rgb1 = _run(self.rgb_layer1, rgb_input)
    rgb2 = _run(self.rgb_layer2, rgb1)
    rgb3 = _run(self.rgb_layer3, rgb2)
    fusion1 = _run(self.fusion_layer1, torch.cat((depth1, rgb1), 1))
    fusion2 = _run(self.fusion_layer2, torch.cat((depth2, rgb2), 1))
    fusion3 = _run(self.fusion_layer3, torch.cat((depth3, rgb3), 1))
    depth5 = _run(self.depth_layer5, torch.cat((fusion3, depth4), 1))
    depth6 = _run(self.depth_layer6, torch.cat((fusion2, depth5), 1))
    depth7 = _run(self.depth_layer7, torch.cat((fusion1, depth6), 1))
    return depth7 * self.maxdepth


class depth_regression(nn.Module):
  """mode_fusion.py:255-265."""

  def __init__(self, maxdepth):
    super(depth_regression, self).__init__()
    self.lastconv = nn.Sequential(convbn(64, 32, 3, 1, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(32, 1, kernel_size=1, padding=0, stride=1, bias=True),
                                  nn.Sigmoid())
    self.maxdepth = torch.tensor(maxdepth)

  def forward(self, x):
    return _run(self.lastconv, x) * self.maxdepth


def _init(model):
  """mode_fusion.py:273-281 / 291-299: He-normal for Conv2d (ConvTranspose2d keeps torch's default), BN to (1, 0)."""
  for m in model.modules():
    if isinstance(m, nn.Conv2d):
      n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
      m.weight.data.normal_(0, math.sqrt(2. / n))
    elif isinstance(m, nn.BatchNorm2d):
      m.weight.data.fill_(1)
      m.bias.data.zero_()
    elif isinstance(m, nn.Linear):
      m.bias.data.zero_()


class Baseline(nn.Module):
  """Depth-only baseline: forward(depthes) with `depthes` a list of (B, 1, H, W) maps (mode_fusion.py:269-283)."""

  def __init__(self, maxdepth):
    super(Baseline, self).__init__()
    self.feature_extraction = feature_extraction_Baseline(maxdepth)
    _init(self)

  def forward(self, depthes):
    return self.feature_extraction(torch.cat(depthes, 1))


class ModeFusion(nn.Module):
  """forward(depthes, confs, rgbs): lists of (B,1,H,W) depth maps, (B,1,H,W) confidence maps and (B,3,H,W) images; depth and
  confidence maps are interleaved channel-wise (mode_fusion.py:301-313)."""

  def __init__(self, maxdepth, channels, inplanes):
    super(ModeFusion, self).__init__()
    self.feature_extraction = feature_extraction_MODE_Fusion(maxdepth, channels, inplanes)
    _init(self)

  def forward(self, depthes, confs, rgbs):
    depthes_confs = []
    for d, c in zip(depthes, confs):
      depthes_confs += [d, c]
    return self.feature_extraction(torch.cat(depthes_confs, 1), torch.cat(rgbs, 1))
