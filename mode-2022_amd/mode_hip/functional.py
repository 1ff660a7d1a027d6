"""torch.autograd wrappers over the C-ABI kernels (host-side glue only; no arithmetic here)."""
import torch

from . import check, lib, profiling, ptr, require_f32c, require_gpu, stream_of


# ------------------------------------------------------------------------------------ cost volume
def cost_volume_fwd(ref, tgt, d4):
  """(B,C,H,W) x2 -> (B,2C,D4,H,W).  Replaces the loop at models/mode_disparity.py:104-113."""
  require_gpu(ref, tgt)
  ref, tgt = ref.contiguous(), tgt.contiguous()
  require_f32c(ref, tgt)
  if ref.shape != tgt.shape or ref.dim() != 4:
    raise ValueError('cost_volume: feature maps must both be (B,C,H,W), got %s and %s' % (tuple(ref.shape), tuple(tgt.shape)))
  B, C, H, W = ref.shape
  cost = torch.empty((B, 2 * C, d4, H, W), dtype=ref.dtype, device=ref.device)
  nbytes = 4 * (2 * ref.numel() + cost.numel())
  with torch.cuda.device_of(ref), profiling.region('cost_volume_fwd', nbytes, 0, ref.device):
    check(lib().mode_cost_volume_fwd(ptr(ref), ptr(tgt), ptr(cost), B, C, d4, H, W, stream_of(ref)), 'mode_cost_volume_fwd')
  return cost


def cost_volume_bwd(gcost, C):
  require_gpu(gcost)
  gcost = gcost.contiguous()
  require_f32c(gcost)
  B, C2, D4, H, W = gcost.shape
  g_ref = torch.empty((B, C, H, W), dtype=gcost.dtype, device=gcost.device)
  g_tgt = torch.empty_like(g_ref)
  nbytes = 4 * (gcost.numel() + 2 * g_ref.numel())
  with torch.cuda.device_of(gcost), profiling.region('cost_volume_bwd', nbytes, 0, gcost.device):
    check(lib().mode_cost_volume_bwd(ptr(gcost), ptr(g_ref), ptr(g_tgt), B, C, D4, H, W, stream_of(gcost)),
          'mode_cost_volume_bwd')
  return g_ref, g_tgt


class CostVolumeFunction(torch.autograd.Function):

  @staticmethod
  def forward(ctx, ref, tgt, d4):
    ctx.C = ref.shape[1]
    return cost_volume_fwd(ref, tgt, d4)

  @staticmethod
  def backward(ctx, gcost):
    g_ref, g_tgt = cost_volume_bwd(gcost, ctx.C)
    return g_ref, g_tgt, None


def cost_volume(ref, tgt, d4):
  return CostVolumeFunction.apply(ref, tgt, d4)


# ------------------------------------------------------------------------------------ sphere conv
def _sc_dims(x_shape, w_shape, out_hw, stride, groups):
  B, Ci, H, W = x_shape
  Co, Cig, Kh, Kw = w_shape
  if Ci != Cig * groups:
    raise RuntimeError('Input shape and kernel channels wont match: (%d vs %d).' % (Ci, Cig * groups))
  return [B, Ci, H, W, Co, Kh, Kw, stride[0], stride[1], out_hw[0], out_hw[1], groups]


def _wpack(w, groups):
  Co, Cig, Kh, Kw = w.shape
  n = lib().mode_sphere_conv_wpack_bytes(Cig * groups, Co, Kh, Kw, groups)
  if n == 0:
    check(-1, 'mode_sphere_conv_wpack_bytes')
  return torch.empty(n // 4, dtype=torch.float32, device=w.device)


def _check_pos(pos, x, Kh, Kw):
  if pos.dim() != 4 or pos.shape[1] != 2 * Kh * Kw:
    raise RuntimeError('invalid number of channels of position')
  if pos.shape[2] != x.shape[2] or pos.shape[3] != x.shape[3]:
    raise RuntimeError('invalid spatial size of position, expected height: %d, width: %d, BUT got height: %d, width: %d' %
                       (x.shape[2], x.shape[3], pos.shape[2], pos.shape[3]))


def sphere_conv_fwd(x, pos, w, out, stride, groups):
  """Writes `out` (B,Co,Ho,Wo) in place.  Replaces sphere_conv_forward_cuda (sphere_conv_cuda.cpp:129-210)."""
  require_gpu(x, pos, w, out)
  require_f32c(x, pos, w, out)
  _check_pos(pos, x, w.shape[2], w.shape[3])
  dims = _sc_dims(x.shape, w.shape, out.shape[2:], stride, groups)
  flops = 2 * out.numel() * w[0].numel()
  nbytes = 4 * (x.numel() + out.numel() + pos.numel() + w.numel())
  with torch.cuda.device_of(x), profiling.region('sphere_conv_fwd', nbytes, flops, x.device):
    wp = _wpack(w, groups)
    check(lib().mode_sphere_conv_fwd(ptr(x), ptr(pos), ptr(w), ptr(out), ptr(wp), *dims, stream_of(x)), 'mode_sphere_conv_fwd')
  return out


def sphere_conv_bwd_data(gy, pos, w, gx, stride, groups):
  """Accumulates into `gx` (B,Ci,H,W) (caller zero-fills, sphere_conv.py:62)."""
  require_gpu(gy, pos, w, gx)
  require_f32c(gy, pos, w, gx)
  dims = _sc_dims(gx.shape, w.shape, gy.shape[2:], stride, groups)
  flops = 2 * gy.numel() * w[0].numel()
  nbytes = 4 * (gx.numel() + gy.numel() + pos.numel() + w.numel())
  with torch.cuda.device_of(gy), profiling.region('sphere_conv_bwd_data', nbytes, flops, gy.device):
    wp = _wpack(w, groups)
    check(lib().mode_sphere_conv_bwd_data(ptr(gy), ptr(pos), ptr(w), ptr(gx), ptr(wp), *dims, stream_of(gy)),
          'mode_sphere_conv_bwd_data')
  return gx


def sphere_conv_bwd_weight(gy, pos, x, gw, stride, groups):
  """Accumulates into `gw` (Co,Ci/g,Kh,Kw) (caller zero-fills, sphere_conv.py:63)."""
  require_gpu(gy, pos, x, gw)
  require_f32c(gy, pos, x, gw)
  dims = _sc_dims(x.shape, gw.shape, gy.shape[2:], stride, groups)
  B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, G = dims
  flops = 2 * gy.numel() * gw[0].numel()
  nbytes = 4 * (x.numel() + gy.numel() + pos.numel() + gw.numel())
  with torch.cuda.device_of(gy), profiling.region('sphere_conv_bwd_weight', nbytes, flops, gy.device):
    n = lib().mode_sphere_conv_bwd_weight_workspace_bytes(B, Ci, Co, Kh, Kw, Ho, Wo, G)
    ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=gy.device)
    check(lib().mode_sphere_conv_bwd_weight(ptr(gy), ptr(pos), ptr(x), ptr(gw), ptr(ws), *dims, stream_of(gy)),
          'mode_sphere_conv_bwd_weight')
  return gw
