"""3D regulariser + soft-argmin head building blocks used by ModeDisparity.forward.

Each helper takes the nn.Module that owns the parameters (so the state_dict layout stays the reference's) and
runs the layer on libmode_hip.so: fp32-MFMA kernels for every 3x3x3 layer of the regulariser (stride-1 and stride-2 Conv3d,
ConvTranspose3d k3 s2 p1 op1, the 32->1 classifier convolutions; forward and both gradients), the fused BatchNorm (+ residual
add) (+ ReLU) kernels, and the fused soft-argmin head.  There is no CPU path and no backend switch: layers whose shape the
kernels do not implement run as the torch module they are (vendor library), everything else always runs on the HIP kernels.
The plain-torch composition of the head, for A/B measurements and the CPU wiring tests, lives in tests/plain_ops.py.
"""
import contextlib
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from mode_hip import functional as HF



def _hip_kind(conv, x):
  """'conv1' / 'conv2' / 'deconv' if libmode_hip implements this layer, else None (vendor library)."""
  if conv.kernel_size != (3, 3, 3) or conv.padding != (1, 1, 1) or conv.dilation != (1, 1, 1) or \
      conv.groups != 1 or conv.bias is not None or conv.in_channels > 64 or conv.out_channels > 64:
    return None
  if type(conv) is nn.Conv3d:
    if conv.stride == (1, 1, 1):
      return 'conv1'
    if conv.stride == (2, 2, 2) and all(s % 2 == 0 for s in x.shape[2:]):
      return 'conv2'
  if type(conv) is nn.ConvTranspose3d and conv.stride == (2, 2, 2) and conv.output_padding == (1, 1, 1):
    return 'deconv'
  return None


def conv3(conv, x, carrier=None):
  """One Conv3d / ConvTranspose3d layer.  carrier: the HF.GradCarrier of x when x has exactly one other consumer (honoured by the 3-D
  convolutions; every other layer kind leaves it unarmed, and autograd adds the two gradients as usual)."""
  if not x.is_cuda:
    raise NotImplementedError('Only support cuda tensor!')  # same refusal as the reference's native op
  kind = _hip_kind(conv, x)
  if kind == 'conv1':
    return HF.conv3d(x, conv.weight, 1, carrier)
  if kind == 'conv2':
    return HF.conv3d(x, conv.weight, 2, carrier)
  if kind == 'deconv':
    return HF.deconv3d(x, conv.weight)
  if type(conv) is nn.Conv2d and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.groups == 1 and \
      conv.bias is None and conv.padding == conv.dilation and conv.dilation in ((1, 1), (2, 2)) and conv.padding_mode == 'zeros' and \
      x.dtype == torch.float32 and x.is_cuda:
    # regular stride-1 3x3 layer of the 2-D extractor: own kernels (csrc/conv2d.hip, conv2d_wgrad.hip)
    if torch.is_grad_enabled() and (conv.weight.requires_grad or x.requires_grad):
      if HF.conv2d_wgrad_supported(x, conv.weight):
        return HF.conv2d_3x3(x.contiguous(), conv.weight, conv.dilation[0], carrier)
    elif HF._conv2d_own(x, conv.weight):
      return HF.conv2d_fwd(x.contiguous(), conv.weight.detach().contiguous(), conv.dilation[0])
  if type(conv) is nn.Conv2d and HF.conv2d_3x3_s2_supported(x, conv) and torch.is_grad_enabled() and (conv.weight.requires_grad or x.requires_grad):
    return HF.conv2d_3x3_s2(x, conv)  # layer2[0].conv1: gradients on the stride-1 MFMA kernels (zero-inserted output gradient)
  if type(conv) is nn.Conv2d and HF.conv_stem_supported(x, conv):
    return HF.conv_stem(x, conv)  # firstconv[0]: 7x7 stride 2 on the image (csrc/conv_stem.hip)
  if type(conv) is nn.Conv2d and HF.conv1x1_supported(x, conv):
    return HF.conv1x1(x, conv)  # downsample branches, lastconv[0] / [4]: plain MFMA GEMMs over the planes (csrc/conv1x1.hip)
  if type(conv) is nn.Conv2d and HF.conv2d_tabled_supported(x, conv):
    # every other regular convolution (7x7 stride 2, 3x3 stride 2, 1x1): the gather-and-MAC kernels on an integer table
    return HF.conv2d_tabled(x, conv)
  return conv(x)


def conv_bn(seq, x, relu=False, add=None, x_carrier=None, add_carrier=None):
  """seq = Sequential(Conv2d | SphereConv | Conv3d | ConvTranspose3d, BatchNorm): y = bn(conv(x)) [+ add] [relu]
  (convbn / convbn_3d / sphereConvbn submodule.py:15-22, 61-74; transposed form mode_disparity.py:23, 25).
  Inference (eval mode, no autograd): ONE launch per layer -- the BatchNorm is folded into the convolution kernel (scale into the
  packed weights, shift / residual / ReLU into its store), so an eval forward contains no BatchNorm launch at all."""
  conv, bn = seq[0], seq[1]
  if x.is_cuda and x.dtype == torch.float32 and HF.bn_foldable(bn):
    y = _conv_bn_folded(conv, bn, x, add, relu)
    if y is not None:
      return y
  if (x.is_cuda and bn.training and current_bn_groups() == 1 and _hip_kind(conv, x) == 'conv1' and conv.out_channels > 1 and
      torch.is_grad_enabled() and HF.conv3d_stats_supported(x, conv.weight, bn)):
    # training, stride-1 3-D layer on the split kernel: the BatchNorm statistics come out of the convolution's epilogue
    return HF.conv3d_bn_train(x, conv.weight, bn, add, relu)
  # x_carrier / add_carrier: gradient carriers of x and of the skip tensor (HF.GradCarrier; training only, None otherwise)
  y = conv3(conv, x) if x_carrier is None else conv3(conv, x, x_carrier)
  return bn_act(bn, y, add, relu) if add_carrier is None else bn_act(bn, y, add, relu, add_carrier)


def _conv_bn_folded(conv, bn, x, add, relu):
  """The layer as one fused launch, or None when no fused form exists for it (then: convolution, then the fused BatchNorm pass)."""
  kind = _hip_kind(conv, x)
  if kind == 'conv1' and conv.out_channels > 1:
    return HF.conv3d_bn_eval(x, conv.weight, bn, 1, add, relu)
  if kind == 'conv2':
    return HF.conv3d_bn_eval(x, conv.weight, bn, 2, add, relu)
  if kind == 'deconv':
    return HF.deconv3d_bn_eval(x, conv.weight, bn, add, relu)
  if hasattr(conv, 'forward_bn'):  # SphereConv
    return conv.forward_bn(x, bn, add, relu)
  if type(conv) is nn.Conv2d and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros':
    if conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == conv.dilation and conv.dilation in ((1, 1), (2, 2)) and \
        HF._conv2d_own(x, conv.weight):
      return HF.conv2d_bn_eval(x, conv.weight, bn, conv.dilation[0], add, relu)
    if HF.conv_stem_supported(x, conv):
      return HF.conv_stem_fwd(x, conv.weight, bn, add, relu)
    if HF.conv1x1_supported(x, conv):
      return HF.conv1x1_fwd(x, conv.weight, conv.stride[0], bn, add, relu)
    if HF.conv2d_tabled_supported(x, conv):
      return HF.conv2d_tabled_bn_eval(x, conv, bn, add, relu)
  return None


# nn.DataParallel (train_disparity.py:264-265 and the other reference call sites) runs forward() of one replica per GPU on
# one Python thread each: the statistics grouping of the current extractor pass is per-thread state, never a module global.
_tls = threading.local()


def current_bn_groups():
  return getattr(_tls, 'bn_groups', 1)


@contextlib.contextmanager
def bn_groups(n):
  """Inside this context every training-mode BatchNorm of THIS THREAD takes its batch statistics per group of B / n consecutive
  samples and updates its running statistics group after group -- exactly what n consecutive calls of the network on the n
  sub-batches do.  ModeDisparity pushes the left and the right images through the shared extractor as one batch this way."""
  prev = current_bn_groups()
  _tls.bn_groups = n
  try:
    yield
  finally:
    _tls.bn_groups = prev


def bn_act(bn, y, add=None, relu=False, add_carrier=None):
  """BatchNorm + optional residual add + optional ReLU: one fused HIP pass (two in training)."""
  if not y.is_cuda:
    raise NotImplementedError('Only support cuda tensor!')
  if not HF.bn_supported(y):  # (other dtypes, B * C beyond the grid limit): the torch module itself, on the GPU
    return bn_act_torch(bn, y, add, relu)
  return HF.bn_act(bn, y, add, relu, groups=current_bn_groups() if bn.training else 1, add_carrier=add_carrier)


def bn_act_torch(bn, y, add=None, relu=False, add_carrier=None):
  """The same layer as separate torch ops; only reached for tensor shapes the fused kernels do not take.  (add_carrier is left unarmed:
  autograd adds the skip's gradient as usual.)"""
  groups = current_bn_groups()
  if groups > 1 and bn.training:  # group after group, like consecutive calls
    y = torch.cat([bn(part) for part in y.chunk(groups, 0)], 0)
  else:
    y = bn(y)
  if add is not None:
    y = y + add
  return F.relu(y, inplace=True) if relu else y


def classify(seq, x, *, add=None, x_carrier=None):
  """classifN = Sequential(convbn_3d, ReLU, Conv3d(32->1)) (mode_disparity.py:76-80).  The keyword-only `add` is the residual the
  reference adds right after the call (`cost2 = classif2(out2) + cost1`, mode_disparity.py:128-129).  Training: BatchNorm + ReLU +
  the single-channel convolution + the add as one operator behind the first convolution (HF.classif_head_train); the activated
  tensor between the two convolutions is never written."""
  conv0, bn = seq[0][0], seq[0][1]
  if (x.is_cuda and bn.training and torch.is_grad_enabled() and current_bn_groups() == 1 and _hip_kind(conv0, x) == 'conv1' and
      conv0.out_channels > 1):
    y = conv3(conv0, x) if x_carrier is None else conv3(conv0, x, x_carrier)
    if HF.classif_fused_supported(y, bn, seq[2]):
      return HF.classif_head_train(y, bn, seq[2], add)
    cost = conv3(seq[2], bn_act(bn, y, None, True))
  else:
    cost = conv3(seq[2], conv_bn(seq[0], x, relu=True) if x_carrier is None else conv_bn(seq[0], x, relu=True, x_carrier=x_carrier))
  return cost if add is None else cost + add


def head(cost, size, with_confidence=False):
  """Trilinear upsample (align_corners=True) of (B,1,D/4,H/4,W/4) logits to `size`=(D,H,W), softmax over D and
  expectation of the disparity index (mode_disparity.py:131-152, submodule.py:50-57); optionally the confidence
  map of mode_disparity.py:157-183 = P(round(d)-1) + P(round(d)) + P(round(d)+1), indices clamped to [0, D-1]."""
  if not cost.is_cuda:
    raise NotImplementedError('Only support cuda tensor!')
  if with_confidence:
    return HF.head_fwd(cost, size, with_confidence=True)
  return HF.head(cost, size)
