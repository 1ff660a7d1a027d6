"""CPU restatement of the reference's native spherical-convolution op.  TEST INFRASTRUCTURE.

Reference (paths relative to the upstream repo):
  models/basic/spherical_conv/src/sphere_conv_cuda_kernel.cu   (K1 im2col :195-262,
      bilinear sampler :83-113, K2 col2im :293-356, gradient weight :128-152)
  models/basic/spherical_conv/src/sphere_conv_cuda.cpp         (forward :129-210,
      backward :213-336 -- per-sample im2col + GEMM over Ci*Kh*Kw, groups)

Two independent formulations are kept so that they can be checked against each other:
  * the *direct* form, a vectorised transcription of the CUDA index arithmetic
    (``im2col`` / ``col2im_scatter``), used as the primary oracle in fp32 and fp64;
  * the *grid_sample* form (``forward_grid_sample``), which expresses the same sampling
    through ``torch.nn.functional.grid_sample(padding_mode='zeros', align_corners=True)``
    and lets autograd derive the backward pass.

No reference test, fixture or golden vector exists for this op: parity unpinned by the
reference's own tests (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F


def out_size(size, k, stride, pad, dil):
  # sphere_conv.py:112-113 / sphere_conv_cuda.cpp:159-162
  return (size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def _tap_coords(pos, k, sH, sW, Ho, Wo):
  """Coordinates of tap k at every output pixel: the table is indexed at (h_out*sH, w_out*sW)
  (cu:221-222, 236-237); channel 2k is the row coordinate, 2k+1 the column coordinate."""
  hs = torch.arange(Ho) * sH
  ws = torch.arange(Wo) * sW
  h = pos[0, 2 * k][hs][:, ws]
  w = pos[0, 2 * k + 1][hs][:, ws]
  return h, w


def im2col(x, pos, Kh, Kw, sH, sW, Ho, Wo):
  """K1, cu:195-262 + cu:83-113.  x (B,C,H,W), pos (1,2*Kh*Kw,H,W) -> col (B,C,Kh*Kw,Ho,Wo).

  Arithmetic is carried out in x.dtype in the operation order of cu:92-111."""
  B, C, H, W = x.shape
  K = Kh * Kw
  pos = pos.to(x.dtype)
  col = x.new_zeros((B, C, K, Ho, Wo))
  for k in range(K):
    h, w = _tap_coords(pos, k, sH, sW, Ho, Wo)
    valid = (h > -1) & (w > -1) & (h < H) & (w < W)  # cu:246
    hl = torch.floor(h).long()
    wl = torch.floor(w).long()
    hh = hl + 1
    wh = wl + 1
    lh = h - hl.to(x.dtype)
    lw = w - wl.to(x.dtype)
    uh = 1 - lh
    uw = 1 - lw

    def corner(hi, wi, ok):
      v = x[:, :, hi.clamp(0, H - 1), wi.clamp(0, W - 1)]
      return torch.where(ok, v, torch.zeros((), dtype=x.dtype))

    v1 = corner(hl, wl, (hl >= 0) & (wl >= 0))
    v2 = corner(hl, wh, (hl >= 0) & (wh <= W - 1))
    v3 = corner(hh, wl, (hh <= H - 1) & (wl >= 0))
    v4 = corner(hh, wh, (hh <= H - 1) & (wh <= W - 1))
    w1, w2, w3, w4 = uh * uw, uh * lw, lh * uw, lh * lw
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    col[:, :, k] = torch.where(valid, val, torch.zeros((), dtype=x.dtype))
  return col


def forward(x, pos, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
  """sphere_conv_forward_cuda, cpp:129-210 (bias path omitted: never used)."""
  B, C, H, W = x.shape
  Co, Cig, Kh, Kw = weight.shape
  Ho = out_size(H, Kh, stride[0], padding[0], dilation[0])
  Wo = out_size(W, Kw, stride[1], padding[1], dilation[1])
  col = im2col(x, pos, Kh, Kw, stride[0], stride[1], Ho, Wo)  # (B,C,K,Ho,Wo)
  col = col.reshape(B, groups, Cig * Kh * Kw, Ho * Wo)
  wm = weight.reshape(groups, Co // groups, Cig * Kh * Kw)
  y = torch.einsum('gok,bgkn->bgon', wm, col)
  return y.reshape(B, Co, Ho, Wo)


def _gradient_weight(ah, aw, h, w, H, W):
  """get_gradient_weight, cu:128-152 (vectorised)."""
  dt = ah.dtype
  outside = (ah <= -1) | (ah >= H) | (aw <= -1) | (aw >= W)
  hl = torch.floor(ah).long()
  wl = torch.floor(aw).long()
  hh = hl + 1
  wh = wl + 1
  hf = h.to(dt)
  wf = w.to(dt)
  weight = torch.zeros_like(ah)
  weight = torch.where((h == hl) & (w == wl), (hf + 1 - ah) * (wf + 1 - aw), weight)
  weight = torch.where((h == hl) & (w == wh), (hf + 1 - ah) * (aw + 1 - wf), weight)
  weight = torch.where((h == hh) & (w == wl), (ah + 1 - hf) * (wf + 1 - aw), weight)
  weight = torch.where((h == hh) & (w == wh), (ah + 1 - hf) * (aw + 1 - wf), weight)
  return torch.where(outside, torch.zeros_like(ah), weight)


def col2im_scatter(gcol, pos, H, W, Kh, Kw, sH, sW):
  """K2, cu:293-356.  gcol (B,C,K,Ho,Wo) -> gx (B,C,H,W); the atomicAdd scatter is an
  index_put_(accumulate=True)."""
  B, C, K, Ho, Wo = gcol.shape
  dt = gcol.dtype
  pos = pos.to(dt)
  gx = gcol.new_zeros((B, C, H * W))
  for k in range(K):
    h, w = _tap_coords(pos, k, sH, sW, Ho, Wo)
    ch = h.to(torch.int64)  # (int) cast = truncation toward zero, cu:337-338
    cw = w.to(torch.int64)
    for dy in (0, 1):
      for dx in (0, 1):
        nh = ch + dy
        nw = cw + dx
        ok = (nh >= 0) & (nh < H) & (nw >= 0) & (nw < W) & \
             ((h - nh.to(dt)).abs() < 1) & ((w - nw.to(dt)).abs() < 1)  # cu:343-346
        gwt = _gradient_weight(h, w, nh, nw, H, W)
        gwt = torch.where(ok, gwt, torch.zeros_like(gwt))
        idx = (nh.clamp(0, H - 1) * W + nw.clamp(0, W - 1)).reshape(-1)
        contrib = (gcol[:, :, k] * gwt).reshape(B, C, Ho * Wo)
        gx.index_add_(2, idx, contrib)
  return gx.reshape(B, C, H, W)


def backward(x, pos, weight, gy, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
  """sphere_conv_backward_cuda, cpp:213-336: returns (grad_input, grad_weight)."""
  B, C, H, W = x.shape
  Co, Cig, Kh, Kw = weight.shape
  Ho, Wo = gy.shape[2:]
  Kd = Cig * Kh * Kw
  wm = weight.reshape(groups, Co // groups, Kd)
  gym = gy.reshape(B, groups, Co // groups, Ho * Wo)
  # cpp:281-284  columns = W^T . gO
  gcol = torch.einsum('gok,bgon->bgkn', wm, gym).reshape(B, C, Kh * Kw, Ho, Wo)
  gx = col2im_scatter(gcol, pos, H, W, Kh, Kw, stride[0], stride[1])  # cpp:291-294
  # cpp:298-315  gW += gO . col^T, accumulated over the batch
  col = im2col(x, pos, Kh, Kw, stride[0], stride[1], Ho, Wo).reshape(B, groups, Kd, Ho * Wo)
  gw = torch.einsum('bgon,bgkn->gok', gym, col).reshape(Co, Cig, Kh, Kw)
  return gx, gw


class _DirectFn(torch.autograd.Function):
  """autograd wrapper so that the full-model oracle back-propagates through the direct form."""

  @staticmethod
  def forward(ctx, x, pos, weight, stride, padding, dilation, groups):
    ctx.save_for_backward(x, pos, weight)
    ctx.cfg = (stride, padding, dilation, groups)
    return forward(x, pos, weight, stride, padding, dilation, groups)

  @staticmethod
  def backward(ctx, gy):
    x, pos, weight = ctx.saved_tensors
    gx, gw = backward(x, pos, weight, gy.contiguous(), *ctx.cfg)
    return gx, None, gw, None, None, None, None


def _pair(v):
  return (v, v) if isinstance(v, int) else tuple(v)


def sphere_conv(x, pos, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
  """Same call signature as the reference's module-global ``sphere_conv``
  (= SphereConvFunction.apply, sphere_conv.py:18, :117), runnable on CPU tensors."""
  y = _DirectFn.apply(x, pos, weight, _pair(stride), _pair(padding), _pair(dilation), groups)
  if bias is not None:
    y = y + bias.view(1, -1, 1, 1)
  return y


def forward_grid_sample(x, pos, weight, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1):
  """Independent formulation: bilinear sampling with per-corner zero padding is
  grid_sample(zeros, align_corners=True) on pixel coordinates; the validity guard of cu:246
  is implied because a coordinate outside (-1, size) has all four corners out of range."""
  B, C, H, W = x.shape
  Co, Cig, Kh, Kw = weight.shape
  Ho = out_size(H, Kh, stride[0], padding[0], dilation[0])
  Wo = out_size(W, Kw, stride[1], padding[1], dilation[1])
  pos = pos.to(x.dtype)
  cols = []
  for k in range(Kh * Kw):
    h, w = _tap_coords(pos, k, stride[0], stride[1], Ho, Wo)
    gy_ = h / (H - 1) * 2 - 1
    gx_ = w / (W - 1) * 2 - 1
    grid = torch.stack((gx_, gy_), -1).unsqueeze(0).expand(B, Ho, Wo, 2)
    cols.append(F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True))
  col = torch.stack(cols, 2).reshape(B, groups, Cig * Kh * Kw, Ho * Wo)
  wm = weight.reshape(groups, Co // groups, Cig * Kh * Kw)
  return torch.einsum('gok,bgkn->bgon', wm, col).reshape(B, Co, Ho, Wo)
