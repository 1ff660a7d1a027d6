"""Spherical convolution operator -- Python side of the operator seam.

Mirrors the public surface of the reference's models/basic/spherical_conv/sphere_conv.py
(``SphereConv`` :120-246, ``SphereConvFunction`` :16-114, module-global ``sphere_conv`` :117) so that
models and scripts written against it run unchanged; the arithmetic happens in libmode_hip.so
(gfx950 kernels) through ``sphere_conv_cuda`` -- the same two-function native seam the reference uses.
"""
import math
import threading

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from . import sphere_conv_cuda
from mode_hip import functional as _F


def _conv_out(size, k, stride, pad, dil):
  return (size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


class SphereConvFunction(Function):
  """autograd glue; same argument list and error behaviour as sphere_conv.py:16-114."""

  @staticmethod
  def forward(ctx, input, position, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, training=False):
    # (training: an extension, set by SphereConv.forward when gradients are being recorded -- ctx.needs_input_grad is also set under
    # no_grad -- and passed on to sphere_conv_forward_cuda: the arithmetic of the training step)
    if input is not None and input.dim() != 4:
      raise ValueError('Expected 4D tensor as input, got {}D tensor instead.'.format(input.dim()))
    ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
    ctx.groups = groups
    ctx.has_bias = bias is not None
    if not input.is_cuda:
      raise NotImplementedError('Only support cuda tensor!')
    if not ctx.has_bias:
      bias = input.new_empty(1)  # placeholder, as in the reference
    input = input.contiguous()
    weight = weight.contiguous()
    output = input.new_empty(SphereConvFunction._infer_shape(ctx, input, weight))
    ctx.save_for_backward(input, position, weight, bias)
    kh, kw = weight.shape[2:]
    # The windowed kernels work on a plane-transposed copy of the input; the weight gradient needs the same copy again, so
    # it is kept with the graph (33.5 MB per layer at the benchmark shape) instead of being rebuilt in the backward pass.
    keep = [] if ctx.needs_input_grad[2] else None  # only the weight gradient uses it
    sphere_conv_cuda.sphere_conv_forward_cuda(input, weight, bias, None, position, output, None, kh, kw, ctx.stride[0],
                                              ctx.stride[1], ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1],
                                              groups, ctx.has_bias, keep_transposed=keep, training=bool(training))
    ctx.input_t = keep[0] if keep else None
    return output

  @staticmethod
  @once_differentiable
  def backward(ctx, grad_output):
    input, position, weight, bias = ctx.saved_tensors
    if not grad_output.is_cuda:
      raise NotImplementedError
    grad_input = torch.empty_like(input)  # written, not added to (overwrite_grad_input)
    # The op ADDS the weight gradient to the tensor it is given (reference contract).  When the parameter has a gradient
    # sink (mode_hip.functional.grad_sink) that tensor is the optimizer's gradient buffer itself and autograd gets None.
    sink = _F.grad_sink(weight) if ctx.needs_input_grad[2] else None
    grad_weight = sink if sink is not None else torch.zeros_like(weight)
    grad_bias = torch.zeros_like(bias)
    kh, kw = weight.shape[2:]
    sphere_conv_cuda.sphere_conv_backward_cuda(input, weight, bias, None, position, None, grad_input, grad_weight, grad_bias,
                                               grad_output, kh, kw, ctx.stride[0], ctx.stride[1], ctx.padding[0], ctx.padding[1],
                                               ctx.dilation[0], ctx.dilation[1], ctx.groups, ctx.has_bias,
                                               overwrite_grad_input=True, input_transposed=ctx.input_t)
    return grad_input, None, (None if sink is not None else grad_weight), (grad_bias if ctx.has_bias else None), None, None, None, None, None

  @staticmethod
  def _output_size(input, weight, padding, dilation, stride):
    size = (input.size(0), weight.size(0)) + tuple(
        _conv_out(input.size(d + 2), weight.size(d + 2), stride[d], padding[d], dilation[d]) for d in range(input.dim() - 2))
    if not all(s > 0 for s in size):
      raise ValueError('convolution input is too small (output would be {})'.format('x'.join(map(str, size))))
    return size

  @staticmethod
  def _infer_shape(ctx, input, weight):
    h, w = input.shape[2:4]
    kh, kw = weight.shape[2:4]
    return (input.size(0), weight.size(0), _conv_out(h, kh, ctx.stride[0], ctx.padding[0], ctx.dilation[0]),
            _conv_out(w, kw, ctx.stride[1], ctx.padding[1], ctx.dilation[1]))


sphere_conv = SphereConvFunction.apply


# ---- operator on plane-transposed storage (an extension; the reference has no counterpart) ----------------------------
# The windowed kernels work on (B, C, W, H) storage.  A run of stride-1 spherical layers separated only by layout-agnostic ops
# (BatchNorm, ReLU, residual add, 1x1 convolution) can stay in that storage: one transpose in, one out, instead of four per
# layer and step.  `transposed_io()` switches SphereConv.forward to this form for the modules called inside the block.
_tls = threading.local()


class transposed_io(object):
  """with transposed_io(): SphereConv.forward takes and returns plane-transposed tensors (B, C, W, H)."""

  def __enter__(self):
    self._prev = getattr(_tls, 'transposed', False)
    _tls.transposed = True

  def __exit__(self, *exc):
    _tls.transposed = self._prev


class TransposePlanes(Function):
  """(B, C, H, W) <-> (B, C, W, H) contiguous; its own adjoint."""

  @staticmethod
  def forward(ctx, x):
    return _F.transpose_planes(x.contiguous())

  @staticmethod
  @once_differentiable
  def backward(ctx, g):
    return _F.transpose_planes(g.contiguous())


class SphereConvTransposedFunction(Function):
  """SphereConvFunction for stride 1, 3x3 taps, no bias, on plane-transposed input and output."""

  @staticmethod
  def forward(ctx, input_t, position, weight, groups, training=False):
    input_t = input_t.contiguous()
    weight = weight.contiguous()
    B, _, W, H = input_t.shape
    output_t = input_t.new_empty((B, weight.size(0), W, H))
    keep = []
    _F.sphere_conv_fwd_t(input_t, position, weight, output_t, groups, f16=bool(training), amax_out=keep)  # (a training step: DESIGN 3v)
    ctx.w_amax = keep[0][1] if keep else None  # the weight's maximum buffer: the input gradient reads the same weight
    ctx.save_for_backward(input_t, position, weight)
    ctx.groups = groups
    return output_t

  @staticmethod
  @once_differentiable
  def backward(ctx, grad_output_t):
    input_t, position, weight = ctx.saved_tensors
    gyt = grad_output_t.contiguous()
    grad_input_t = None
    if ctx.needs_input_grad[0]:
      grad_input_t = torch.empty_like(input_t)
      _F.sphere_conv_bwd_data_t(gyt, position, weight, grad_input_t, ctx.groups, w_amax=ctx.w_amax)
    grad_weight = None
    if ctx.needs_input_grad[2]:
      sink = _F.grad_sink(weight)
      gw = sink if sink is not None else torch.zeros_like(weight)
      _F.sphere_conv_bwd_weight_t(gyt, position, input_t, gw, ctx.groups)
      grad_weight = None if sink is not None else gw
    return grad_input_t, None, grad_weight, None, None

# All layers of one network share a handful of geometries (16 identical tables in ModeDisparity): build each
# table once per process and upload it once per device, instead of per layer (reference) and per call (:240).
_table_lock = threading.Lock()
_host_tables = {}
_device_tables = {}


def make_sphere_position(height, width, sphere_type, kernel_size):
  """Sampling table of sphere_conv.py:180-237 for an equirectangular grid of `height` x `width`
  (width == 2*height): float32 (1, 2*Kh*Kw, H, W) for 'ERP', (1, 2*Kh*Kw, W, H) for 'Cassini'.

  Gnomonic (tangent-plane) kernel a la SphereNet.  The reference's float64 operation sequence is kept
  so that the float32 table is bit-identical (tests/test_host.py checks it against golden vectors),
  including its quirks: kerY divides by cos(range_y * delta_lon) (:194), the centre tap uses rho = 1e-8
  (:198-199) and longitudes wrap with `% width` (:225) while latitudes do not.  Latitude depends on the
  row only and longitude is `row-term + column`, so the work is O(H*Kh*Kw) transcendental evaluations plus
  one broadcast add -- the reference evaluates Python list comprehensions over all W columns."""
  Kh, Kw = kernel_size
  d_lat = np.pi / height
  d_lon = 2 * np.pi / width

  def taps(k):
    r = np.arange(-(k // 2), k // 2 + 1)
    return r if k % 2 else np.delete(r, k // 2)

  tx, ty = taps(Kw), taps(Kh)
  ker_x, ker_y = np.meshgrid(np.tan(tx * d_lon), np.tan(ty * d_lat) / np.cos(ty * d_lon))
  rho = np.sqrt(ker_x**2 + ker_y**2)
  if Kh % 2 and Kw % 2:
    rho[Kh // 2][Kw // 2] = 1e-8
  nu = np.arctan(rho)
  cos_nu, sin_nu = np.cos(nu), np.sin(nu)
  lat0 = ((np.arange(0, height) / height) - 0.5) * np.pi
  lon0 = ((np.arange(0, width) / width) - 0.5) * (2 * np.pi)
  s, c = np.sin(lat0).reshape(-1, 1, 1), np.cos(lat0).reshape(-1, 1, 1)
  lat_rows = np.arcsin(cos_nu * s + ker_y * sin_nu * c / rho)  # (H,Kh,Kw)
  dlon_rows = np.arctan2(ker_x * sin_nu, (rho * c * cos_nu - ker_y * s * sin_nu))  # (H,Kh,Kw)
  lat = np.broadcast_to(((lat_rows / np.pi + 0.5) * height)[:, None], (height, width, Kh, Kw))
  lon = dlon_rows[:, None] + lon0[None, :, None, None]
  lon = ((lon / (2 * np.pi) + 0.5) * width) % width
  if sphere_type == 'ERP':
    table = np.stack((lat, lon)).astype(np.float32).transpose(3, 4, 0, 1, 2)  # (Kh,Kw,(lat,lon),H,W)
  else:  # Cassini: transposed grid, (lon, lat) order
    table = np.stack((lon, lat)).astype(np.float32).transpose(3, 4, 0, 2, 1)
  return torch.from_numpy(np.ascontiguousarray(table.reshape(1, 2 * Kh * Kw, *table.shape[3:])))


def _host_table(height, width, sphere_type, kernel_size):
  key = (height, width, sphere_type, tuple(kernel_size))
  with _table_lock:
    if key not in _host_tables:
      _host_tables[key] = make_sphere_position(height, width, sphere_type, kernel_size)
    return _host_tables[key]


class SphereConv(nn.Module):
  """Drop-in for the reference's ``SphereConv`` (sphere_conv.py:120-246): same constructor, attributes,
  parameter names/shapes and initialisation; ``position`` is a plain attribute (not in the state_dict)."""

  def __init__(self, in_height, in_width, sphereType, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
               groups=1, bias=False):
    super(SphereConv, self).__init__()
    assert (sphereType is not None) and (sphereType in ['Cassini', 'ERP'])
    assert (in_height is not None) and (in_height > 0)
    assert (in_width is not None) and (in_width > 0)
    assert in_channels % groups == 0, 'in_channels {} cannot be divisible by groups {}'.format(in_channels, groups)
    assert out_channels % groups == 0, 'out_channels {} cannot be divisible by groups {}'.format(out_channels, groups)
    in_h, in_w = min(in_height, in_width), max(in_height, in_width)
    assert in_w == 2 * in_h

    self.in_height, self.in_width = in_h, in_w
    self.in_channels, self.out_channels = in_channels, out_channels
    self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
    self.padding, self.dilation = _pair(padding), _pair(dilation)
    self.groups = groups
    self.sphereType = sphereType
    self.transposed = False  # nn.Conv2d look-alike attributes, as in the reference
    self.output_padding = _single(0)
    self.input_size = (1, in_channels, in_h, in_w)
    self.output_size = self.cal_output_size()
    self.position = self.gen_sphere_position()
    self.position.requires_grad = False
    self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
    if bias:
      self.bias = nn.Parameter(torch.zeros(out_channels))  # the reference crashes here (nn.parameter typo, :153)
    else:
      self.register_parameter('bias', None)
    self.reset_parameters()

  def reset_parameters(self):
    stdv = 1. / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
    self.weight.data.uniform_(-stdv, stdv)

  def cal_output_size(self):
    hw = (self.in_height, self.in_width)
    size = (1, self.out_channels) + tuple(
        _conv_out(hw[d], self.kernel_size[d], self.stride[d], self.padding[d], self.dilation[d]) for d in range(2))
    if not all(s > 0 for s in size):
      raise ValueError('convolution input is too small (output would be {})'.format('x'.join(map(str, size))))
    return size

  def gen_sphere_position(self):
    return _host_table(self.in_height, self.in_width, self.sphereType, self.kernel_size)

  def position_on(self, device):
    """The table on `device` (uploaded once per device and geometry, shared by all layers and replicas)."""
    if self.position.device == device:
      return self.position
    key = (self.in_height, self.in_width, self.sphereType, self.kernel_size, str(device))
    with _table_lock:
      t = _device_tables.get(key)
      if t is None:
        t = _device_tables[key] = self.position.to(device)
    return t

  def forward(self, x):
    training = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)  # gradients are being recorded
    if getattr(_tls, 'transposed', False):
      if not self.supports_transposed_io(x.shape[0], x.device):
        raise RuntimeError('SphereConv: this layer cannot run on plane-transposed storage (see supports_transposed_io)')
      return SphereConvTransposedFunction.apply(x, self.position_on(x.device), self.weight, self.groups, training)
    return sphere_conv(x, self.position_on(x.device), self.weight, self.bias, self.stride, self.padding, self.dilation,
                       self.groups, training)

  def forward_bn(self, x, bn, add=None, relu=False):
    """Inference only (an extension; the reference has no counterpart): relu?(bn(self(x)) [+ add]) with `bn` in eval mode as ONE
    launch -- the BatchNorm folded into the convolution kernel (models/stage3d.conv_bn).  None if this layer has no fused form."""
    if self.bias is not None or self.dilation != (1, 1):
      return None
    pos = self.position_on(x.device)
    if getattr(_tls, 'transposed', False):
      if not self.supports_transposed_io(x.shape[0], x.device):
        raise RuntimeError('SphereConv: this layer cannot run on plane-transposed storage (see supports_transposed_io)')
      return _F.sphere_conv_bn_eval(x, pos, self.weight, bn, self.stride, self.groups, add, relu, transposed=True)
    if SphereConvFunction._infer_shape(self, x, self.weight)[2:] != ((x.shape[2] - 1) // self.stride[0] + 1,
                                                                     (x.shape[3] - 1) // self.stride[1] + 1):
      return None  # (not the 'same' geometry the fused entry assumes)
    return _F.sphere_conv_bn_eval(x, pos, self.weight, bn, self.stride, self.groups, add, relu)

  def supports_transposed_io(self, batch, device):
    """Whether forward() can run inside `with transposed_io()` for this batch size: stride 1, 3x3, no bias, a sampling table
    the windowed kernels can plan, and enough tiles to fill the chip."""
    if self.stride != (1, 1) or self.kernel_size != (3, 3) or self.bias is not None or device.type != 'cuda':
      return False
    return _F.sphere_t_supported(self.position_on(device), self.weight, batch, self.groups)

  def getPosition(self):
    return self.position

  def extra_repr(self):
    return '{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, sphereType={sphereType}'.format(
        **self.__dict__)
