"""Sub-networks of the disparity stage (module tree and state_dict names of the reference's
models/submodule.py; regular 2D convolutions run on the vendor library, spherical ones on libmode_hip)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import stage3d
from .basic import SphereConv
from .basic.spherical_conv import sphere_conv as sphere_conv_mod


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
  """Conv2d (no bias) + BatchNorm2d; a dilated conv pads by its dilation (submodule.py:15-17)."""
  return nn.Sequential(
      nn.Conv2d(in_planes, out_planes, kernel_size, stride, dilation if dilation > 1 else pad, dilation, bias=False),
      nn.BatchNorm2d(out_planes))


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
  """Conv3d (no bias) + BatchNorm3d (submodule.py:20-22)."""
  return nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size, stride, pad, bias=False), nn.BatchNorm3d(out_planes))


def sphereConvbn(in_height, in_width, sphereType, in_planes, out_planes, kernel_size, stride, pad, dilation):
  """SphereConv (no bias) + BatchNorm2d (submodule.py:61-74)."""
  return nn.Sequential(
      SphereConv(in_height, in_width, sphereType, in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                 padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False), nn.BatchNorm2d(out_planes))


def sphereConvbnrelu(in_height, in_width, sphereType, in_planes, out_planes, kernel_size, stride, pad, dilation):
  seq = sphereConvbn(in_height, in_width, sphereType, in_planes, out_planes, kernel_size, stride, pad, dilation)
  seq.add_module('2', nn.ReLU(inplace=True))
  return seq


class disparityregression(nn.Module):
  """Expectation over the disparity axis (submodule.py:50-57).  Kept for API compatibility; ModeDisparity
  itself uses the fused head."""

  def __init__(self, maxdisp):
    super(disparityregression, self).__init__()
    self.maxdisp = maxdisp

  def forward(self, x):
    disp = torch.arange(self.maxdisp, dtype=x.dtype, device=x.device).view(1, self.maxdisp, 1, 1)
    return torch.sum(x * disp, 1, keepdim=True)


def _run_convbn_relu_chain(seq, x):
  """Sequential(convbn, ReLU, convbn, ReLU, ...) with every BatchNorm+ReLU as one fused pass."""
  mods = list(seq)
  i = 0
  while i < len(mods):
    relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
    if isinstance(mods[i], nn.Sequential) and len(mods[i]) == 2 and isinstance(mods[i][1], nn.BatchNorm2d):
      x = stage3d.conv_bn(mods[i], x, relu=relu)
      i += 2 if relu else 1
    else:  # anything else (a bare Conv2d, pooling, ...) runs as is
      x = mods[i](x)
      i += 1
  return x


class _ResidualBlock(nn.Module):
  expansion = 1
  input_shared = False  # True: the block's input has consumers outside the block (plain attribute, not state)

  def _residual(self, x, final_relu):
    """conv-bn-relu, conv-bn, + shortcut [, relu]: the second BatchNorm, the add and the ReLU are one fused pass
    (the reference runs bn, `out += x` and relu as three kernels)."""
    # identity skip: x has two consumers, conv1 and the add behind conv2's BatchNorm -- the skip's gradient (which the BatchNorm backward
    # produces first) is added inside conv1's input-gradient kernel instead of by a pass of autograd's (HF.GradCarrier; regular 3x3
    # layers on the split kernels -- a spherical conv1 leaves the carrier unarmed and nothing changes)
    # (not when x is also used outside the block -- `input_shared`, set by the extractor for layer3[0], whose input is concatenated into
    # the output as well: with three consumers the carrier would change the association of autograd's sum)
    car = stage3d.HF.grad_carrier(x) if (self.downsample is None and not self.input_shared and isinstance(self.conv1[0][0], nn.Conv2d)) else None
    if car is not None:
      out = stage3d.conv_bn(self.conv1[0], x, relu=True, x_carrier=car)
      return stage3d.conv_bn(self.conv2, out, relu=final_relu, add=x, add_carrier=car)
    out = stage3d.conv_bn(self.conv1[0], x, relu=True)
    shortcut = x if self.downsample is None else stage3d.conv_bn(self.downsample, x)
    return stage3d.conv_bn(self.conv2, out, relu=final_relu, add=shortcut)


class BasicBlock(_ResidualBlock):
  """PSMNet block without the trailing ReLU (submodule.py:25-47); only used by conv='Regular'."""

  def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
    super(BasicBlock, self).__init__()
    self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
    self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
    self.downsample = downsample
    self.stride = stride

  def forward(self, x):
    return self._residual(x, False)


class RegularBasicBlock(_ResidualBlock):
  """conv-bn-relu, conv-bn, + skip, relu (submodule.py:94-119)."""

  def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
    super(RegularBasicBlock, self).__init__()
    self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
    self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
    self.relu = nn.ReLU(inplace=True)
    self.downsample = downsample
    self.stride = stride

  def forward(self, x):
    return self._residual(x, True)


class SphereBasicBlock(_ResidualBlock):
  """Same residual block with spherical convolutions (submodule.py:122-147)."""

  def __init__(self, in_height, in_width, sphereType, inplanes, planes, stride, downsample, pad, dilation):
    super(SphereBasicBlock, self).__init__()
    self.conv1 = nn.Sequential(sphereConvbn(in_height, in_width, sphereType, inplanes, planes, 3, stride, pad, dilation),
                               nn.ReLU(inplace=True))
    self.conv2 = sphereConvbn(in_height // stride, in_width // stride, sphereType, planes, planes, 3, 1, pad, dilation)
    self.relu = nn.ReLU(inplace=True)
    self.downsample = downsample
    self.stride = stride

  def forward(self, x):
    return self._residual(x, True)


def _downsample(inplanes, planes, stride):
  return nn.Sequential(nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(planes))


def _three_convs(cin, first_kernel, first_pad):
  return nn.Sequential(convbn(cin, 32, first_kernel, 2, first_pad, 1), nn.ReLU(inplace=True), convbn(32, 32, 3, 1, 1, 1),
                       nn.ReLU(inplace=True), convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))


class sphere_feature_extraction(nn.Module):
  """2D feature extractor with a spherical-convolution stage (submodule.py:151-201):
  firstconv (7x7 s2 + 2x 3x3) -> layer1 (3 blocks 32->64) -> layer2 (8 blocks, s2) -> layer3 (4 blocks, dil 2)
  -> layer4 (8 sphere blocks 64->128) ; cat(layer2, layer3, layer4) -> lastconv -> 32 ch at 1/4 resolution.

  `transposed_chain` (plain attribute, default on): layer4 runs end to end on the plane-transposed storage of the windowed
  spherical kernels when every layer supports it; the parity tests switch it off per instance to compare with the NCHW operator."""
  transposed_chain = True

  def __init__(self, in_height, in_width, sphereType):
    super(sphere_feature_extraction, self).__init__()
    self.inplanes = 32
    self.firstconv = _three_convs(3, 7, 3)
    h2, w2, h4, w4 = in_height // 2, in_width // 2, in_height // 4, in_width // 4
    self.layer1 = self._make_layer(RegularBasicBlock, h2, w2, sphereType, 32, 64, 3, 1, 1, 1)
    self.layer2 = self._make_layer(RegularBasicBlock, h2, w2, sphereType, 64, 64, 8, 2, 1, 1)
    self.layer3 = self._make_layer(RegularBasicBlock, h4, w4, sphereType, 64, 64, 4, 1, 1, 2)
    self.layer4 = self._make_layer(SphereBasicBlock, h4, w4, sphereType, 64, 128, 8, 1, 1, 1)
    self.layer3[0].input_shared = True  # layer2's output also goes into the concatenation of forward()
    self.lastconv = nn.Sequential(convbn(256, 128, 1, 1, 0, 1), nn.ReLU(inplace=True), convbn(128, 128, 3, 1, 1, 1),
                                  nn.ReLU(inplace=True), convbn(128, 32, 1, 1, 0, 1), nn.ReLU(inplace=True))

  def _make_layer(self, block, height, width, sphereType, inplanes, planes, blocks, stride, pad, dilation):
    down = _downsample(inplanes, planes, stride) if (stride != 1 or inplanes != planes * block.expansion) else None
    sphere = block is SphereBasicBlock
    print("add {} block. num: {}, inplanes: {}, planes: {}".format('sphere' if sphere else 'regular', blocks, inplanes, planes))
    geo = (lambda s: (height // s, width // s, sphereType)) if sphere else (lambda s: ())
    layers = [block(*geo(1), inplanes, planes, stride, down, pad, dilation)]
    layers += [block(*geo(stride), planes * block.expansion, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
    return nn.Sequential(*layers)

  def forward(self, x):
    raw = self.layer2(self.layer1(_run_convbn_relu_chain(self.firstconv, x)))
    regular = self.layer3(raw)
    sphere = self._layer4(regular)
    return _run_convbn_relu_chain(self.lastconv, torch.cat((raw, regular, sphere), 1))

  def _layer4(self, x):
    """The 16 spherical layers.  Between them sit only BatchNorm, ReLU, the residual add and one 1x1 convolution, none of
    which cares about the order of the two spatial axes, so the whole run stays in the plane-transposed storage of the
    windowed kernels (one transpose in, one out) when every layer supports it."""
    convs = [m for m in self.layer4.modules() if isinstance(m, SphereConv)]
    if self.transposed_chain and x.is_cuda and all(m.supports_transposed_io(x.shape[0], x.device) for m in convs):
      with sphere_conv_mod.transposed_io():
        yt = self.layer4(sphere_conv_mod.TransposePlanes.apply(x))
      return sphere_conv_mod.TransposePlanes.apply(yt)
    return self.layer4(x)


class feature_extraction(nn.Module):
  """PSMNet SPP feature extractor of ModeDisparity(conv='Regular') (submodule.py:205-268).  Vendor 2D
  convolutions, poolings and bilinear upsampling; every BatchNorm (+ residual add) (+ ReLU) on the fused HIP kernels, like the
  spherical extractor."""

  def __init__(self):
    super(feature_extraction, self).__init__()
    self.inplanes = 32
    self.firstconv = _three_convs(3, 3, 1)
    self.layer1 = self._make_layer(BasicBlock, 32, 3, 1, 1, 1)
    self.layer2 = self._make_layer(BasicBlock, 64, 16, 2, 1, 1)
    self.layer3 = self._make_layer(BasicBlock, 128, 3, 1, 1, 1)
    self.layer4 = self._make_layer(BasicBlock, 128, 3, 1, 1, 2)
    for i, k in enumerate((64, 32, 16, 8), 1):
      setattr(self, 'branch%d' % i, nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)), convbn(128, 32, 1, 1, 0, 1),
                                                  nn.ReLU(inplace=True)))
    self.lastconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                  nn.Conv2d(128, 32, kernel_size=1, padding=0, stride=1, bias=False))

  def _make_layer(self, block, planes, blocks, stride, pad, dilation):
    down = _downsample(self.inplanes, planes * block.expansion, stride) if (
        stride != 1 or self.inplanes != planes * block.expansion) else None
    layers = [block(self.inplanes, planes, stride, down, pad, dilation)]
    self.inplanes = planes * block.expansion
    layers += [block(self.inplanes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
    return nn.Sequential(*layers)

  def forward(self, x):
    raw = self.layer2(self.layer1(_run_convbn_relu_chain(self.firstconv, x)))
    skip = self.layer4(self.layer3(raw))
    size = skip.shape[2:]
    pooled = [F.interpolate(_run_convbn_relu_chain(getattr(self, 'branch%d' % i), skip), size, mode='bilinear', align_corners=True)
              for i in (4, 3, 2, 1)]
    return _run_convbn_relu_chain(self.lastconv, torch.cat([raw, skip] + pooled, 1))
