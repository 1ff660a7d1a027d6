"""Native seam of the spherical-convolution operator, MI355X edition.

The reference imports a compiled pybind11 module of this name (sphere_conv.py:12) exposing
``sphere_conv_forward_cuda`` / ``sphere_conv_backward_cuda`` (sphere_conv_cuda.cpp:339-345).  This
module exposes the same two callables with the same positional arguments, so code written against
the reference extension keeps working; the work is done by libmode_hip.so (hand-written gfx950
kernels behind a C-ABI, include/mode_hip.h).  The module name is kept for drop-in compatibility only:
there is no CUDA code path.

Argument semantics follow the reference:
  * ``ones`` and ``columns`` are caller-owned scratch tensors that the reference re-allocates
    internally (cpp:164-175); they are accepted and ignored (no column buffer exists here);
  * ``output`` is written in place; ``grad_input`` / ``grad_weight`` / ``grad_bias`` are
    accumulated into and must be zero-filled by the caller (sphere_conv.py:62-64);
  * pad / dilation only take part in the output-size check (cpp:159-162);
  * shape violations raise RuntimeError (TORCH_CHECK / AT_ERROR in cpp:40-126, 152-157).
"""
import torch

from mode_hip import functional as _F


def _out_hw(h, w, kh, kw, sh, sw, ph, pw, dh, dw):
  return ((h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def _shape_check(input, position, grad_output, weight, kH, kW, sH, sW, pH, pW, dH, dW, group):
  """shape_check, sphere_conv_cuda.cpp:40-126."""
  if weight.dim() != 4:
    raise RuntimeError('4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: %d' % weight.dim())
  if not weight.is_contiguous():
    raise RuntimeError('weight tensor has to be contiguous')
  if kW <= 0 or kH <= 0:
    raise RuntimeError('kernel size should be greater than zero, but got kH: %d kW: %d' % (kH, kW))
  if weight.size(2) != kH or weight.size(3) != kW:
    raise RuntimeError('kernel size should be consistent with weight, but got kH: %d kW: %d weight.size(2): %d, '
                       'weight.size(3): %d' % (kH, kW, weight.size(2), weight.size(3)))
  if sW <= 0 or sH <= 0:
    raise RuntimeError('stride should be greater than zero, but got dH: %d dW: %d' % (sH, sW))
  if dW <= 0 or dH <= 0:
    raise RuntimeError('dilation should be greater than 0, but got dilationH: %d dilationW: %d' % (dH, dW))
  if input.dim() != 4:
    raise RuntimeError('3D or 4D input tensor expected but got: %d' % input.dim())
  n_in = weight.size(1) * group
  H, W = input.size(2), input.size(3)
  Ho, Wo = _out_hw(H, W, kH, kW, sH, sW, pH, pW, dH, dW)
  if Ho < 1 or Wo < 1:
    raise RuntimeError('Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). Output size is too small' %
                       (n_in, H, W, weight.size(0), Ho, Wo))
  if input.size(1) != n_in:
    raise RuntimeError('invalid number of input planes, expected: %d, but got: %d' % (n_in, input.size(1)))
  if H < kH or W < kW:
    raise RuntimeError('input image is smaller than kernel')
  if position.size(2) != H or position.size(3) != W:
    raise RuntimeError('invalid spatial size of position, expected height: %d, width: %d, BUT got height: %d, width: %d' %
                       (H, W, position.size(2), position.size(3)))
  if position.size(1) != 2 * kH * kW:
    raise RuntimeError('invalid number of channels of position')
  if grad_output is not None:
    if grad_output.size(1) != weight.size(0):
      raise RuntimeError('invalid number of gradOutput planes, expected: %d, but got: %d' % (weight.size(0), grad_output.size(1)))
    if grad_output.size(2) != Ho or grad_output.size(3) != Wo:
      raise RuntimeError('invalid size of gradOutput, expected height: %d width: %d , but got height: %d width: %d' %
                         (Ho, Wo, grad_output.size(2), grad_output.size(3)))
  return Ho, Wo


def _computes_in_fp32(name, input):
  """The reference dispatches its kernels with AT_DISPATCH_FLOATING_TYPES_AND_HALF (sphere_conv_cuda_kernel.cu:273, 367): double, float
  and half tensors are accepted, anything else raises.  The gfx950 kernels are fp32: float64 and float16 tensors are converted on the
  way in and the results converted back into the caller's buffers (float64 callers therefore get fp32 accuracy -- INTEGRATION.md says
  so; float16 callers get MORE than the reference's half arithmetic).  Returns True when such a conversion is needed."""
  if input.dtype == torch.float32:
    return False
  if input.dtype in (torch.float64, torch.float16):
    return True
  raise RuntimeError('"%s" not implemented for \'%s\'' % (name, str(input.dtype).replace('torch.', '')))  # (the text of AT_DISPATCH's error)


def sphere_conv_forward_cuda(input, weight, bias, ones, position, output, columns, kernel_h, kernel_w, stride_h, stride_w,
                             pad_h, pad_w, dilation_h, dilation_w, group, has_bias, *, keep_transposed=None, training=False):
  """The 17 positional arguments of the reference op.  Keyword-only extension: pass a list as keep_transposed and the
  plane-transposed copy of `input` that the windowed kernel made (if it ran) is appended to it, for
  sphere_conv_backward_cuda(input_transposed=...).  training=True (SphereConvFunction with a gradient to compute) lets the windowed
  kernel use the two-piece fp16 arithmetic of the training step (functional.SPHERE_FWD_F16); an inference call keeps three bf16 pieces."""
  Ho, Wo = _shape_check(input, position, None, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                        dilation_w, group)
  if tuple(output.shape) != (input.size(0), weight.size(0), Ho, Wo):
    raise RuntimeError('output has shape %s, expected %s' % (tuple(output.shape), (input.size(0), weight.size(0), Ho, Wo)))
  if _computes_in_fp32('sphere_conv_forward_cuda', input):
    out32 = torch.empty(output.shape, dtype=torch.float32, device=input.device)
    sphere_conv_forward_cuda(input.float(), weight.float().contiguous(), None, None, position.float(), out32, None, kernel_h, kernel_w,
                             stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, False)
    output.copy_(out32)
    if has_bias:
      output += bias.view(1, -1, 1, 1)
    return
  xt = _F.sphere_conv_fwd(input.contiguous(), position.contiguous(), weight, output, (stride_h, stride_w), group, return_transposed=True,
                          f16=training)
  if keep_transposed is not None and xt is not None:
    keep_transposed.append(xt)
  if has_bias:
    output += bias.view(1, -1, 1, 1)  # cpp:207-209


def sphere_conv_backward_cuda(input, weight, bias, ones, position, columns, grad_input, grad_weight, grad_bias, grad_output,
                              kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, has_bias, *,
                              overwrite_grad_input=False, input_transposed=None):
  """The 20 positional arguments of the reference op.  Keyword-only extensions: overwrite_grad_input=True lets the caller
  pass an uninitialised grad_input (it is written, not added to), which saves the zero-fill and one read of the tensor;
  input_transposed = the plane-transposed copy of `input` kept from the forward (see sphere_conv_forward_cuda)."""
  _shape_check(input, position, grad_output, grad_weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
               dilation_w, group)
  if _computes_in_fp32('sphere_conv_backward_cuda', input):
    gi32 = torch.empty(grad_input.shape, dtype=torch.float32, device=input.device)
    gw32 = torch.zeros(grad_weight.shape, dtype=torch.float32, device=input.device)
    sphere_conv_backward_cuda(input.float(), weight.float().contiguous(), None, None, position.float(), None, gi32, gw32, None,
                              grad_output.float(), kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, False,
                              overwrite_grad_input=True)
    if overwrite_grad_input:
      grad_input.copy_(gi32)
    else:
      grad_input += gi32.to(grad_input.dtype)
    grad_weight += gw32.to(grad_weight.dtype)
    if has_bias:
      grad_bias += grad_output.sum((0, 2, 3))
    return
  gy = grad_output.contiguous()
  pos = position.contiguous()
  # one plane-transposed copy of grad_output serves both gradients (windowed weight gradient, transposed adjoint gather)
  gyt = None
  if ((stride_h, stride_w) == (1, 1) and gy.shape[2:] == input.shape[2:] and
      _F.sphere_uses_transposed_copies(pos, kernel_h, kernel_w)):  # (Cassini-like tables; an ERP table runs on the NCHW tensors as they are)
    gyt = _F.transpose_planes(gy)
  _F.sphere_conv_bwd_data(gy, pos, weight.contiguous(), grad_input, (stride_h, stride_w), group, overwrite=overwrite_grad_input,
                          gy_transposed=gyt)
  _F.sphere_conv_bwd_weight(gy, pos, input.contiguous(), grad_weight, (stride_h, stride_w), group, x_transposed=input_transposed,
                            gy_transposed=gyt)
  if has_bias:
    grad_bias += gy.sum((0, 2, 3))  # cpp:316-322
