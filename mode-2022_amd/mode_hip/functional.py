"""torch.autograd wrappers over the C-ABI kernels (host-side glue: shapes, workspaces, autograd; no arithmetic of the path is left
to torch or to a vendor library)."""
import collections
import ctypes
import threading

import torch

from . import BnEpilogue, check, lib, profiling, ptr, require_f32c, require_gpu, stream_of


def _stream_capturing():
  """True while the calling thread's current stream is being captured into a hipGraph."""
  return torch.cuda.is_initialized() and torch.cuda.is_current_stream_capturing()


class _LRU(object):
  """Bounded cache for the per-geometry device tables (sampling tables of the integer-table convolutions, tile plans, adjoint
  tables, transposed tables): a process that evaluates many resolutions (fisheye / 3D60 / Deep360 in one run) would otherwise pin
  one set per shape for ever -- the adjoint of the 7x7 stem at 1024 x 512 alone is ~100 MB.  Callers hold their own lock.

  Graph safety: a captured hipGraph (mode_hip.graph_step, or any torch.cuda.graph capture) bakes the raw addresses of the tables its
  kernels were launched with into its nodes.  An entry that is handed out WHILE A CAPTURE IS ACTIVE is therefore pinned: it no longer
  counts against maxsize and is never evicted (a later replay would read freed or reused memory and return silently wrong numbers).
  Pins are released only by release_graph_pins(), which callers may use once every graph that used the tables is gone."""
  capturing = staticmethod(_stream_capturing)

  def __init__(self, maxsize):
    self.maxsize = maxsize
    self.d = collections.OrderedDict()
    self.pinned = {}

  def _touch(self, key):
    if key in self.pinned:
      return self.pinned[key]
    value = self.d[key]
    if self.capturing():
      self.pinned[key] = self.d.pop(key)
    else:
      self.d.move_to_end(key)
    return value

  def get(self, key, default=None):
    if key in self.pinned or key in self.d:
      return self._touch(key)
    return default

  def __contains__(self, key):
    return key in self.pinned or key in self.d

  def __getitem__(self, key):
    return self._touch(key)

  def __setitem__(self, key, value):
    if key in self.pinned or self.capturing():
      self.d.pop(key, None)
      self.pinned[key] = value
      return
    self.d[key] = value
    self.d.move_to_end(key)
    while len(self.d) > self.maxsize:
      self.d.popitem(last=False)

  def __len__(self):
    return len(self.d) + len(self.pinned)

  def release_graph_pins(self):
    """Make the pinned entries evictable again (they re-enter the LRU order as the most recent ones)."""
    for key, value in self.pinned.items():
      self.d[key] = value
    self.pinned.clear()
    while len(self.d) > self.maxsize:
      self.d.popitem(last=False)


TABLE_CACHE_ENTRIES = 24  # per cache; one ModeDisparity geometry uses 1 plan, 2 adjoints, 1 transposed table and 5 integer tables


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
# ------------------------------------------------------------------------------------ regular 3x3 Conv2d: own weight gradient
def conv2d_bwd_weight(gy, x, dilation=1, into=None):
  """Weight gradient (Co, Ci, 3, 3) of a stride-1 3x3 convolution with padding = dilation (1 or 2): mode_conv2d_bwd_weight.
  `into`: add to this tensor (a gradient sink) instead of returning a new one."""
  require_gpu(gy, x)
  require_f32c(gy, x)
  B, Ci, H, W = x.shape
  Co = gy.shape[1]
  assert tuple(gy.shape) == (B, Co, H, W)
  gw = into if into is not None else torch.empty((Co, Ci, 3, 3), dtype=x.dtype, device=x.device)
  flops = 2 * gy.numel() * Ci * 9
  nbytes = 4 * (gy.numel() + x.numel() + gw.numel())
  with torch.cuda.device_of(x), profiling.region('conv2d_bwd_weight[%d->%d d%d %dx%d]' % (Ci, Co, dilation, H, W) if profiling.ENABLED
                                                 else 'conv2d_bwd_weight', nbytes, flops, x.device):
    ws = torch.empty(max(lib().mode_conv2d_bwd_weight_workspace_bytes(B, Ci, H, W, Co) // 4, 1), dtype=torch.float32, device=x.device)
    entry = 'mode_conv2d_bwd_weight_split' if CONV_ARITH == 'bf16x6' else 'mode_conv2d_bwd_weight'  # (any channel counts: masked blocks)
    am = ()
    if _conv2d_f16(True):  # (a weight gradient is a training step: the two-piece fp16 arithmetic)
      entry, am = entry + '_f16', (ptr(_tagged_abs_max(gy)), ptr(_tagged_abs_max(x)))
    check(getattr(lib(), entry)(ptr(gy), ptr(x), *am, ptr(gw), ptr(ws), B, Ci, H, W, Co, dilation, 1 if into is not None else 0, stream_of(x)),
          entry)
  return gw


CONV2D_OWN_MAX_PIXELS = 1024 * 512  # own forward / input gradient up to this H*W


def _conv2d_own(x, w):
  """Whether forward / input gradient of this layer run on mode_conv2d_fwd / _bwd_data.  Measured at the step's 4 images
  (tools/microbench.py --only conv2d): 90-130 TFLOP/s against the vendor's 80-112 at every shape of the extractor, half
  resolution included, and the fusion network's full-resolution layers (1024 x 512: 32.2 -> 31.9 ms per training step); larger
  images stay on the vendor library (untested territory for the tile choice)."""
  co, ci = int(w.shape[0]), int(w.shape[1])
  # the fp32 kernels take up to 128 channels on either side; wider layers (the fusion network's 256-channel bottleneck) need the split
  # kernels in both directions (forward: reduction ci, input gradient: reduction co -- multiples of 16)
  wide_ok = CONV_ARITH == 'bf16x6' and ci % 16 == 0 and co % 16 == 0 and max(ci, co) <= 512
  return (x.shape[2] * x.shape[3] <= CONV2D_OWN_MAX_PIXELS and ((co <= 128 and ci <= 128) or wide_ok) and
          max(co, ci) * x.shape[2] * x.shape[3] < 2**29)


def conv2d_wgrad_supported(x, w):
  """mode_conv2d_bwd_weight addresses a sample with 32-bit lane offsets (csrc/conv2d_wgrad.hip: max(Ci, Co) * H * W < 2^29)."""
  return max(w.shape[0], w.shape[1]) * x.shape[2] * x.shape[3] < 2**29


CONV2D_F16 = True  # forward and input gradient of the 3 x 3 layers of a TRAINING step on two fp16 pieces / three MFMAs per product


def _conv2d_f16(f16):
  return bool(f16) and CONV2D_F16 and CONV_ARITH == 'bf16x6'


def _conv2d_run(entry, name, src, w, out_channels, dilation, f16=False, w_amax=None, amax_out=None):
  require_gpu(src, w)
  require_f32c(src, w)
  B, _, H, W = src.shape
  Co, Ci = w.shape[:2]
  out = torch.empty((B, out_channels, H, W), dtype=src.dtype, device=src.device)
  flops = 2 * B * H * W * Ci * Co * 9
  nbytes = 4 * (src.numel() + out.numel() + w.numel())
  with torch.cuda.device_of(src), profiling.region('%s[%d->%d d%d %dx%d]' % (name, Ci, Co, dilation, H, W) if profiling.ENABLED else name,
                                                   nbytes, flops, src.device):
    wp = torch.empty(lib().mode_conv2d_wpack_bytes(Ci, Co) // 4, dtype=torch.float32, device=src.device)
    which = int(entry == 'mode_conv2d_bwd_data')
    if CONV_ARITH == 'bf16x6' and lib().mode_conv2d_split_supported(Ci, Co, dilation, which) == 1:
      if _conv2d_f16(f16):
        aw = _weight_abs_max(w, w_amax)
        if amax_out is not None:
          amax_out.append(aw)
        am = (ptr(_tagged_abs_max(src)), ptr(aw))
        if which:
          check(lib().mode_conv2d_bwd_data_split_f16(ptr(src), ptr(w), am[0], am[1], None, ptr(out), ptr(wp), B, Ci, H, W, Co, dilation,
                                                     stream_of(src)), 'mode_conv2d_bwd_data_split_f16')
        else:
          check(lib().mode_conv2d_fwd_split_f16(ptr(src), ptr(w), am[0], am[1], ptr(out), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(src)),
                'mode_conv2d_fwd_split_f16')
      elif which:
        check(lib().mode_conv2d_bwd_data_split(ptr(src), ptr(w), ptr(out), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(src)),
              'mode_conv2d_bwd_data_split')
      else:
        check(lib().mode_conv2d_fwd_split(ptr(src), ptr(w), None, ptr(out), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(src)),
              'mode_conv2d_fwd_split')
    else:
      check(getattr(lib(), entry)(ptr(src), ptr(w), ptr(out), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(src)), entry)
  return out


def conv2d_fwd(x, w, dilation=1, f16=False, amax_out=None):
  """f16: the caller is a training step (Conv2d3x3Function with gradients being recorded): the split kernel may use the fp16 arithmetic;
  amax_out: a list that receives the weight's maximum buffer when it did (for the layer's backward)."""
  return _conv2d_run('mode_conv2d_fwd', 'conv2d_fwd', x, w, w.shape[0], dilation, f16, amax_out=amax_out)


def conv2d_bwd_data(gy, w, dilation=1, acc=None, f16=True, w_amax=None):
  """acc: a gradient of the same tensor that is already there; the sum is returned (added in the split kernel's store where it runs).
  (A backward pass is a training step: the fp16 arithmetic where CONV2D_F16 says so; w_amax: the weight's maximum from the forward.)"""
  if acc is None:
    return _conv2d_run('mode_conv2d_bwd_data', 'conv2d_bwd_data', gy, w, w.shape[1], dilation, f16, w_amax=w_amax)
  require_gpu(gy, w, acc)
  acc = acc.contiguous()
  require_f32c(gy, w, acc)
  B, _, H, W = gy.shape
  Co, Ci = w.shape[:2]
  if not (CONV_ARITH == 'bf16x6' and lib().mode_conv2d_split_supported(Ci, Co, dilation, 1) == 1 and tuple(acc.shape) == (B, Ci, H, W)):
    return _conv2d_run('mode_conv2d_bwd_data', 'conv2d_bwd_data', gy, w, Ci, dilation, f16, w_amax=w_amax).add_(acc)
  gx = torch.empty((B, Ci, H, W), dtype=gy.dtype, device=gy.device)
  with torch.cuda.device_of(gy), profiling.region('conv2d_bwd_data[%d->%d d%d %dx%d]' % (Ci, Co, dilation, H, W) if profiling.ENABLED else 'conv2d_bwd_data',
                                                  4 * (gy.numel() + 2 * gx.numel() + w.numel()), 2 * B * H * W * Ci * Co * 9, gy.device):
    wp = torch.empty(lib().mode_conv2d_wpack_bytes(Ci, Co) // 4, dtype=torch.float32, device=gy.device)
    if _conv2d_f16(f16):
      check(lib().mode_conv2d_bwd_data_split_f16(ptr(gy), ptr(w), ptr(_tagged_abs_max(gy)), ptr(_weight_abs_max(w, w_amax)), ptr(acc), ptr(gx), ptr(wp),
                                                 B, Ci, H, W, Co, dilation, stream_of(gy)), 'mode_conv2d_bwd_data_split_f16')
    else:
      check(lib().mode_conv2d_bwd_data_split_acc(ptr(gy), ptr(w), ptr(acc), ptr(gx), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(gy)),
            'mode_conv2d_bwd_data_split_acc')
  return gx


class Conv2d3x3Function(torch.autograd.Function):
  """y = conv2d(x, w, stride 1, padding = dilation).  Weight gradient always on mode_conv2d_bwd_weight (the vendor's runs as an
  NHWC implicit GEMM between two layout transposes); forward and input gradient on mode_conv2d_fwd / _bwd_data where they are
  faster (_conv2d_own), else on the vendor library's fp32 Winograd."""

  @staticmethod
  def forward(ctx, x, w, dilation, carrier=None, training=False):
    ctx.save_for_backward(x, w)
    ctx.dilation = dilation
    ctx.carrier = carrier  # GradCarrier of x (x has one other consumer: the skip of its residual block), or None
    ctx.own = _conv2d_own(x, w)
    ctx.w_amax = None  # the weight's maximum buffer, when the forward ran on the fp16 arithmetic: the input gradient reads the same weight
    if ctx.own:
      keep = []
      y = conv2d_fwd(x, w.contiguous(), dilation, f16=training, amax_out=keep)  # (training: gradients are being recorded -- set by conv2d_3x3)
      ctx.w_amax = keep[0] if keep else None
      return y
    return torch.nn.functional.conv2d(x, w, None, 1, dilation, dilation)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    dil = ctx.dilation
    gy = gy.contiguous()
    gx = None
    if ctx.needs_input_grad[0]:
      prev = ctx.carrier.take(ctx) if ctx.carrier is not None else None  # the skip's gradient, when it came first (it always does)
      if ctx.own:
        gx = conv2d_bwd_data(gy, w.contiguous(), dil, acc=prev, w_amax=ctx.w_amax)
      else:
        gx = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [True, False, False])[0]
        if prev is not None:
          gx = gx.add_(prev)
      if prev is None and ctx.carrier is not None and ctx.carrier.leave(gx, ctx):
        gx = None
    gw = None
    if ctx.needs_input_grad[1]:
      sink = grad_sink(w)
      gw = conv2d_bwd_weight(gy, x.contiguous(), dil, into=sink)
      if sink is not None:
        gw = None
    return gx, gw, None, None, None


def conv2d_3x3(x, w, dilation=1, carrier=None):
  """Callers check conv2d_wgrad_supported(x, w) first (models/stage3d.conv3 does)."""
  if carrier is not None:
    carrier.arm(x.requires_grad and torch.is_grad_enabled())
  return Conv2d3x3Function.apply(x, w, dilation, carrier, torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))


# ------------------------------------------------------------------------------------ cost volume + dres0[0][0], fused
class CostConvAssemble(torch.autograd.Function):
  """out (B,Co,D,H,W) from the partial products R, T (B, 9*Co, H, W) -- see cost_conv() and csrc/cost_conv.hip."""

  @staticmethod
  def forward(ctx, R, T, D):
    require_gpu(R, T)
    require_f32c(R, T)
    B, C9, H, W = R.shape
    Co = C9 // 9
    out = torch.empty((B, Co, D, H, W), dtype=R.dtype, device=R.device)
    nbytes = 4 * (R.numel() + T.numel() + out.numel())
    with torch.cuda.device_of(R), profiling.region('cost_conv_assemble_fwd', nbytes, 0, R.device):
      check(lib().mode_cost_conv_assemble_fwd(ptr(R), ptr(T), ptr(out), B, Co, D, H, W, stream_of(R)), 'mode_cost_conv_assemble_fwd')
    ctx.D = D
    return out

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gout):
    gout = gout.contiguous()
    B, Co, D, H, W = gout.shape
    gR = torch.empty((B, 9 * Co, H, W), dtype=gout.dtype, device=gout.device)
    gT = torch.empty_like(gR)
    nbytes = 4 * (gR.numel() + gT.numel() + gout.numel())
    with torch.cuda.device_of(gout), profiling.region('cost_conv_assemble_bwd', nbytes, 0, gout.device):
      check(lib().mode_cost_conv_assemble_bwd(ptr(gout), ptr(gR), ptr(gT), B, Co, D, H, W, stream_of(gout)), 'mode_cost_conv_assemble_bwd')
    return gR, gT, None


def _tap_products(fea, wpart):
  """(B, 9*Co, H, W): channel (kd*3+kw)*Co + o = sum_{c,kh} wpart[o,c,kd,kh,kw] * fea[b,c,h+kh-1,w] (zero padding in h): a
  convolution with a 3 x 1 kernel from C to 9*Co channels -- the gather-and-MAC kernels on an integer table (conv2d_tabled: own
  forward, input gradient and weight gradient; autograd carries the weight gradient back through the permutation)."""
  Co, C = wpart.shape[:2]
  wr = wpart.permute(2, 4, 0, 1, 3).reshape(9 * Co, C, 3, 1)          # rows (kd, kw, o), taps kh
  return Conv2dTabledFunction.apply(fea, wr, (1, 1), (1, 0), (1, 1))


def cost_conv_supported(fea, d4, co):
  """Limits of the two assembly kernels (csrc/cost_conv.hip): the adjoint keeps the D4 gradient rows of one (sample, channel,
  image row) in LDS, the forward the 18 partial-product rows; beyond that the caller builds the volume (mode_cost_volume_fwd)
  and runs the 64 -> 32 convolution on it."""
  B, _, H, W = fea.shape
  return 4 * (4 + d4 * (W + 4)) <= 160 * 1024 and 4 * (9 * (W + 2) + 9 * W) <= 160 * 1024 and B * co * H < 2**31


def cost_conv(ref, tgt, weight, d4):
  """conv3d(cost_volume(ref, tgt, d4), weight, stride 1, padding 1) for a (Co, 2C, 3, 3, 3) weight, without the volume:
  models/mode_disparity.py:104-116.  261 GFLOP per sample of the reference layer become 7 GFLOP of GEMM plus one HBM-bound
  assembly pass; differentiable in ref, tgt and weight (the GEMMs through autograd, the assembly through its adjoint kernel)."""
  C = ref.shape[1]
  assert tuple(weight.shape[1:]) == (2 * C, 3, 3, 3) and ref.shape == tgt.shape
  return CostConvAssemble.apply(_tap_products(ref, weight[:, :C]), _tap_products(tgt, weight[:, C:]), d4)


def _sc_dims(x_shape, w_shape, out_hw, stride, groups):
  B, Ci, H, W = x_shape
  Co, Cig, Kh, Kw = w_shape
  if Ci != Cig * groups:
    raise RuntimeError('Input shape and kernel channels wont match: (%d vs %d).' % (Ci, Cig * groups))
  return [B, Ci, H, W, Co, Kh, Kw, stride[0], stride[1], out_hw[0], out_hw[1], groups]


def _tag2(name, w, x):
  """Profiling label with the layer shape, e.g. sphere_conv_fwd[128->128 256x128]."""
  return '%s[%d->%d %dx%d]' % (name, x.shape[1], w.shape[0], x.shape[2], x.shape[3]) if profiling.ENABLED else name


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


_plan_cache = _LRU(TABLE_CACHE_ENTRIES)
_plan_lock = threading.Lock()
SPHERE_LAYOUT = 'transposed'  # windowed kernels on plane-transposed copies | 'nchw'
SPHERE_FWD_MIN_WG = 200  # fewer workgroups than this: the windowed forward under-fills the chip, use the general kernel
SPHERE_FWD = 'window'  # 'window' (LDS-window kernels where the table allows) | 'gather'


def transpose_planes(t, out=None):
  """(B, C, H, W) -> (B, C, W, H) contiguous (or the inverse into `out`, whose last two sizes are swapped)."""
  require_gpu(t)
  require_f32c(t)
  B, C, H, W = t.shape
  if out is None:
    out = torch.empty((B, C, W, H), dtype=t.dtype, device=t.device)
  with torch.cuda.device_of(t):
    check(lib().mode_transpose_planes(ptr(t), ptr(out), B * C, H, W, stream_of(t)), 'mode_transpose_planes')
  return out


def sphere_plan(pos, kh, kw):
  """Tile plan of the windowed kernels for the sampling table `pos`: (tiles int32 device tensor, (n_small, n_mid, n_wrap),
  pixels of the non-small tiles int32 device tensor, their number, record weights, record offsets, polar items or None) or None when the table is not spatially compact (the
  general gather kernels are used then).  Built on the host by
  mode_sphere_plan_build once per (table, device) and cached, like the adjoint table."""
  key = (pos.data_ptr(), pos._version, tuple(pos.shape), kh, kw, str(pos.device))
  with _plan_lock:
    if key in _plan_cache:
      return _plan_cache[key][0]
    H, W = pos.shape[2:]
    host = pos.detach().to('cpu', torch.float32).contiguous()
    n = lib().mode_sphere_plan_max_tiles(H, W)
    tiles = torch.empty(4 * n, dtype=torch.int32)
    counts = torch.zeros(4, dtype=torch.int32)
    check(lib().mode_sphere_plan_build(ptr(host), H, W, kh, kw, ptr(tiles), ptr(counts)), 'mode_sphere_plan_build')
    c = [int(v) for v in counts]
    plan = None
    if not c[3]:
      rest = torch.empty(H * W, dtype=torch.int32)
      nrest = torch.zeros(1, dtype=torch.int32)
      check(lib().mode_sphere_plan_rest_pixels(ptr(tiles), ptr(counts), H, W, ptr(rest), ptr(nrest)), 'mode_sphere_plan_rest_pixels')
      nrest = int(nrest)
      nrec = max(lib().mode_sphere_plan_records_count(c[0]), 1)
      rec_w = torch.zeros(4 * nrec, dtype=torch.float32)
      rec_off = torch.zeros(nrec, dtype=torch.int32)
      if c[0]:
        check(lib().mode_sphere_plan_records(ptr(host), ptr(tiles), ptr(counts), H, W, ptr(rec_w), ptr(rec_off)), 'mode_sphere_plan_records')
      polar = None
      npmax = lib().mode_sphere_plan_polar_max_items(ptr(counts))
      if npmax and c[0]:
        pitems = torch.zeros(20 * npmax, dtype=torch.int32)
        prw = torch.zeros(4 * 288 * npmax, dtype=torch.float32)
        pro = torch.zeros(288 * npmax, dtype=torch.int32)
        npol = torch.zeros(1, dtype=torch.int32)
        check(lib().mode_sphere_plan_polar(ptr(host), ptr(tiles), ptr(counts), H, W, ptr(pitems), ptr(prw), ptr(pro), ptr(npol)),
              'mode_sphere_plan_polar')
        if int(npol) > 0:
          n = int(npol)
          polar = (pitems[:20 * n].contiguous().to(pos.device), prw[:4 * 288 * n].contiguous().to(pos.device),
                   pro[:288 * n].contiguous().to(pos.device), n)
      plan = (tiles.to(pos.device), (c[0], c[1], c[2]), rest[:max(nrest, 1)].contiguous().to(pos.device), nrest, rec_w.to(pos.device),
              rec_off.to(pos.device), polar)
    _plan_cache[key] = (plan, pos)  # keep `pos` alive: the key uses its address
    return plan


SPHERE_FWD_F16 = True  # the small-window tiles of a TRAINING forward on two fp16 pieces / three MFMAs per product (bf16x6 only)


SPHERE_BWD_F16 = True  # ... and the windowed input gradient (the adjoint on the same kernel structure)


def _tagged_abs_max(t):
  """The maximum buffer of an ACTIVATION or GRADIENT t: the producer's tag (BatchNorm passes), an earlier call's, or a pass -- whose result
  is left as the tag (known_abs_max drops it when t is written; a tensor that lives inside one step)."""
  am = known_abs_max(t)
  if am is None:
    am = abs_max(t)
    t._mode_amax = (am, t._version, t.data_ptr())
  return am


def _weight_abs_max(w, given=None):
  """The maximum buffer of a WEIGHT: the one the layer's forward computed when the caller kept it (`given`: autograd functions carry it from
  their forward to their backward, where the weight is the same tensor by autograd's own rules), else the entry of the forward pass's
  batched table (weight_maxima: all of a model's weights in one launch at the top of its forward), else a pass.  Never cached on the
  parameter: a write through `.data` moves no version counter, and a stale maximum of a weight that grew is an fp16 overflow -- the
  table lives for ONE forward pass and is computed inside it."""
  if given is not None:
    return given
  table = getattr(_wmax_tls, 'table', None)
  if table is not None:
    hit = table.get(w.data_ptr())
    if hit is not None and hit[1] == w._version and hit[2] == w.numel():
      return hit[0]
  return abs_max(w)


_wmax_tls = threading.local()


class weight_maxima(object):
  """with weight_maxima(module): ...   Inside, the maximum buffers of ALL convolution weights of `module` come from ONE launch
  (mode_abs_max_batch: one workgroup per tensor) issued on entry, instead of a zero fill + a pass per layer -- 63 + 63 launches of ~4.5 us
  per training step of ModeDisparity (0.6 ms of 55).  Only while the fp16 arithmetic is on and gradients are being recorded; the
  table of device pointers is kept on the module (rebuilt when a parameter's storage moved) so that the launch is graph-capturable."""

  def __init__(self, module):
    self.module = module
    self.prev = None

  def __enter__(self):
    self.prev = getattr(_wmax_tls, 'table', None)
    m = self.module
    on = CONV_ARITH == 'bf16x6' and (CONV3D_S1_F16 or SPHERE_FWD_F16 or CONV2D_F16) and torch.is_grad_enabled() and m.training
    ws = [p for p in m.parameters() if p.dim() >= 4 and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()] if on else []
    if not ws:
      _wmax_tls.table = None
      return self
    key = tuple(p.data_ptr() for p in ws)
    cache = m.__dict__.get('_mode_wmax_cache')
    if cache is None or cache[0] != key:
      dev = ws[0].device
      cache = (key, torch.tensor(key, dtype=torch.int64).to(dev), torch.tensor([p.numel() for p in ws], dtype=torch.int64).to(dev))
      m.__dict__['_mode_wmax_cache'] = cache
    out = torch.empty((len(ws), BN_ABSMAX_FLOATS), dtype=torch.float32, device=ws[0].device)
    with torch.cuda.device_of(out), profiling.region('abs_max_batch', 4 * sum(p.numel() for p in ws), 0, out.device):
      check(lib().mode_abs_max_batch(ptr(cache[1]), ptr(cache[2]), len(ws), ptr(out), stream_of(out)), 'mode_abs_max_batch')
    _wmax_tls.table = {p.data_ptr(): (out[i], p._version, p.numel()) for i, p in enumerate(ws)}
    return self

  def __exit__(self, *exc):
    _wmax_tls.table = self.prev
    return False


def _sphere_f16_maxima(x, w, f16):
  """(max |x|, max |w|) buffers for the fp16 arithmetic of the windowed forward, or None when it does not apply."""
  if not (f16 and SPHERE_FWD_F16 and CONV_ARITH == 'bf16x6' and w.shape[1] % 16 == 0):  # (16 input channels per MFMA: the split kernel's layers)
    return None
  return (_tagged_abs_max(x), _weight_abs_max(w))


def sphere_conv_fwd(x, pos, w, out, stride, groups, return_transposed=False, f16=False):
  """Writes `out` (B,Co,Ho,Wo) in place.  Replaces sphere_conv_forward_cuda (sphere_conv_cuda.cpp:129-210).
  return_transposed: return the plane-transposed copy of x the windowed kernel used (None if it did not run) instead of out.
  f16: the caller is a training step (SphereConvFunction with a gradient to compute): the windowed kernel may use the fp16 arithmetic."""
  require_gpu(x, pos, w, out)
  require_f32c(x, pos, w, out)
  _check_pos(pos, x, w.shape[2], w.shape[3])
  dims = _sc_dims(x.shape, w.shape, out.shape[2:], stride, groups)
  flops = 2 * out.numel() * w[0].numel()
  nbytes = 4 * (x.numel() + out.numel() + pos.numel() + w.numel())
  plan = None
  if SPHERE_FWD == 'window' and tuple(stride) == (1, 1) and w.shape[2] * w.shape[3] == 9 and tuple(out.shape[2:]) == tuple(x.shape[2:]):
    plan = sphere_plan(pos, w.shape[2], w.shape[3])
    if plan is not None:
      # one workgroup per 64x4 tile, sample and 128-channel slice: below ~one workgroup per CU the general kernel (64-pixel
      # tiles) fills the chip better (measured at B = 1, 256x128: 22.9 vs 20.3 ms for the whole eval forward)
      n_wg = sum(plan[1]) * x.shape[0] * groups * (-(-(w.shape[0] // groups) // 128))
      if n_wg < SPHERE_FWD_MIN_WG:
        plan = None
    if not _plan_usable(plan):
      post = sphere_native_t(pos, w.shape[2], w.shape[3])  # ERP-like table: the `_t` operator on the NCHW tensors themselves
      if post is not None and sphere_t_supported(post, w, x.shape[0], groups):
        sphere_conv_fwd_t(x, post, w, out, groups, f16=f16)
        return None if return_transposed else out
  with torch.cuda.device_of(x), profiling.region(_tag2('sphere_conv_fwd', w, x), nbytes, flops, x.device):
    if plan is not None:
      B, Ci, H, W, Co, Kh, Kw = dims[:7]
      tiles, (n0, n1, n2) = plan[:2]
      wp = torch.empty(lib().mode_sphere_conv_win_wpack_bytes(Ci, Co, Kh, Kw, groups) // 4, dtype=torch.float32, device=w.device)
      xt = None
      amax = _sphere_f16_maxima(x, w, f16)
      if SPHERE_LAYOUT == 'transposed':
        xt, yt = transpose_planes(x), torch.empty((B, Co, W, H), dtype=x.dtype, device=x.device)
        _sphere_fwd_win(ptr(xt), pos, w, None, ptr(yt), wp, tiles, n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups, 1, stream_of(x), amax)
        transpose_planes(yt, out)
      else:
        _sphere_fwd_win(ptr(x), pos, w, None, ptr(out), wp, tiles, n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups, 0, stream_of(x), amax)
    else:
      xt = None
      wp = _wpack(w, groups)
      check(lib().mode_sphere_conv_fwd(ptr(x), ptr(pos), ptr(w), ptr(out), ptr(wp), *dims, stream_of(x)), 'mode_sphere_conv_fwd')
  return xt if return_transposed else out


def _sphere_fwd_win(xp, pos, w, e, yp, wp, tiles, n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream, amax=None):
  """Windowed spherical forward (optional folded-BatchNorm epilogue `e`): the small-window tiles on the split-bf16 kernel when
  CONV_ARITH says so (the library falls back to the fp32 kernels by itself when the channel count does not fit).
  amax = (max |x|, max |w|) device scalars: the fp16 arithmetic (_sphere_f16_maxima; no epilogue)."""
  if amax is not None and e is None and CONV_ARITH == 'bf16x6':
    check(lib().mode_sphere_conv_fwd_win_split_f16(xp, ptr(pos), ptr(w), ptr(amax[0]), ptr(amax[1]), yp, ptr(wp), ptr(tiles), n0, n1, n2, B, Ci, H,
                                                   W, Co, Kh, Kw, groups, transposed, stream), 'mode_sphere_conv_fwd_win_split_f16')
  elif CONV_ARITH == 'bf16x6':
    check(lib().mode_sphere_conv_fwd_win_split(xp, ptr(pos), ptr(w), ctypes.byref(e) if e is not None else None, yp, ptr(wp), ptr(tiles), n0, n1,
                                               n2, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream), 'mode_sphere_conv_fwd_win_split')
  elif e is not None:
    check(lib().mode_sphere_conv_fwd_win_bn(xp, ptr(pos), ptr(w), ctypes.byref(e), yp, ptr(wp), ptr(tiles), n0, n1, n2, B, Ci, H, W, Co, Kh, Kw,
                                            groups, transposed, stream), 'mode_sphere_conv_fwd_win_bn')
  else:
    check(lib().mode_sphere_conv_fwd_win(xp, ptr(pos), ptr(w), yp, ptr(wp), ptr(tiles), n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups,
                                         transposed, stream), 'mode_sphere_conv_fwd_win')


_adjoint_cache = _LRU(TABLE_CACHE_ENTRIES)
_adjoint_lock = threading.Lock()
SPHERE_BWD_DATA = 'gather'  # 'gather' (adjoint table) | 'scatter' (atomics)


def sphere_adjoint(pos, kh, kw, stride, out_hw):
  """Device copies (rowptr int32, entries int32 pairs) of the transposed sampling table of `pos`; built on the host by
  mode_sphere_adjoint_build once per (table, stride, output size, device) and cached -- the table is a constant of the
  module (sphere_conv.py:150)."""
  key = (pos.data_ptr(), pos._version, tuple(pos.shape), kh, kw, tuple(stride), tuple(out_hw), str(pos.device))
  with _adjoint_lock:
    hit = _adjoint_cache.get(key)
    if hit is not None:
      return hit
    H, W = pos.shape[2:]
    host = pos.detach().to('cpu', torch.float32).contiguous()
    nmax = lib().mode_sphere_adjoint_max_entries(kh, kw, out_hw[0], out_hw[1])
    rowptr = torch.empty(kh * kw * H * W + 1, dtype=torch.int32)
    entries = torch.empty(2 * nmax, dtype=torch.int32)
    n = ctypes.c_int64(0)
    check(lib().mode_sphere_adjoint_build(ptr(host), H, W, kh, kw, stride[0], stride[1], out_hw[0], out_hw[1], ptr(rowptr),
                                          ptr(entries), ctypes.cast(ctypes.pointer(n), ctypes.c_void_p)), 'mode_sphere_adjoint_build')
    hit = (rowptr.to(pos.device), entries[:2 * max(n.value, 1)].contiguous().to(pos.device), pos)  # keep `pos` alive: key uses its address
    _adjoint_cache[key] = hit
    return hit


SPHERE_BWD_DATA_T = True  # adjoint gather on plane-transposed storage
_pos_t_cache = _LRU(TABLE_CACHE_ENTRIES)


def _transposed_table(pos):
  """The sampling table of the plane-transposed problem: image (W rows, H columns), row coordinate = the original column
  coordinate and vice versa.  Cached per table like the adjoint."""
  key = (pos.data_ptr(), pos._version, tuple(pos.shape), str(pos.device))
  with _adjoint_lock:
    hit = _pos_t_cache.get(key)
    if hit is None:
      K2, H, W = pos.shape[1:]
      p = pos[0].view(K2 // 2, 2, H, W)
      t = torch.stack((p[:, 1].transpose(1, 2), p[:, 0].transpose(1, 2)), 1).reshape(1, K2, W, H).contiguous()
      hit = _pos_t_cache[key] = (t, pos)
    return hit[0]


def _plan_usable(plan):
  """A plan the windowed forward / weight-gradient kernels can run as a whole: compact tiles exist and the tall-window tiles (if
  any) have polar items."""
  return plan is not None and plan[1][0] > 0 and not (plan[3] and not (plan[6] is not None and SPHERE_POLAR))


def sphere_native_t(pos, kh, kw):
  """sphereType = 'ERP' on the fast path.  The windowed kernels want the shift-invariant (longitude) axis of the table along the
  lanes and contiguous in memory.  For the Cassini layout that axis is H, the strided one, hence the plane-transposed copies.  For
  an ERP table it is W -- already contiguous: an ERP problem on NCHW storage IS the Cassini problem of the transposed table on
  plane-transposed storage (sphere_conv.py:226-236 builds the Cassini table as exactly that transpose), so the `_t` operators run
  on the NCHW tensors as they are, with no transpose at all.  Returns the transposed table when `pos` itself cannot be planned but
  its transpose can; else None."""
  if kh * kw != 9 or SPHERE_LAYOUT != 'transposed' or _plan_usable(sphere_plan(pos, kh, kw)):
    return None
  post = _transposed_table(pos)
  return post if _plan_usable(sphere_plan(post, kh, kw)) else None


def sphere_uses_transposed_copies(pos, kh, kw):
  """Whether the gradients of this table run on plane-transposed COPIES of their operands (Cassini-like tables); ERP-like tables
  (sphere_native_t) and tables without a plan do not."""
  return SPHERE_LAYOUT == 'transposed' and kh * kw == 9 and _plan_usable(sphere_plan(pos, kh, kw))


def sphere_conv_bwd_data(gy, pos, w, gx, stride, groups, overwrite=False, gy_transposed=None):
  """Accumulates into `gx` (B,Ci,H,W) (caller zero-fills, sphere_conv.py:62); with overwrite=True `gx` may hold anything and
  is overwritten.  gy_transposed: the plane-transposed copy of gy, if the caller has one (the weight gradient needs it too):
  with overwrite=True the gather then runs on the transposed problem, where the 64 lanes of a wave are 64 consecutive rows
  of one column and the source pixels of a tap are runs of consecutive addresses -- on NCHW storage they are one address
  per image row (0.50 vs 0.66 ms per 128->128 layer and 4 images, including the transpose of the result)."""
  require_gpu(gy, pos, w, gx)
  require_f32c(gy, pos, w, gx)
  dims = _sc_dims(gx.shape, w.shape, gy.shape[2:], stride, groups)
  flops = 2 * gy.numel() * w[0].numel()
  nbytes = 4 * (gx.numel() + gy.numel() + pos.numel() + w.numel())
  if SPHERE_BWD_DATA == 'gather':
    B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, G = dims
    post = sphere_native_t(pos, Kh, Kw) if ((sH, sW) == (1, 1) and (Ho, Wo) == (H, W) and SPHERE_BWD_DATA_T) else None
    if post is not None:  # ERP-like table: the adjoint gather of the transposed problem reads and writes the NCHW tensors directly
      if overwrite:
        return sphere_conv_bwd_data_t(gy, post, w, gx, groups)
      gx += sphere_conv_bwd_data_t(gy, post, w, torch.empty_like(gx), groups)
      return gx
    use_t = (SPHERE_BWD_DATA_T and overwrite and gy_transposed is not None and (sH, sW) == (1, 1) and (Ho, Wo) == (H, W) and
             tuple(gy_transposed.shape) == (B, Co, Wo, Ho))
    if use_t:  # the operator on the plane-transposed problem (windowed split kernel where the adjoint plan allows), then back
      gxt = sphere_conv_bwd_data_t(gy_transposed, pos, w, torch.empty((B, Ci, W, H), dtype=gx.dtype, device=gx.device), groups)
      with torch.cuda.device_of(gy):
        transpose_planes(gxt, gx)
      return gx
    with torch.cuda.device_of(gy), profiling.region(_tag2('sphere_conv_bwd_data', w, gx), nbytes, flops, gy.device):
      wp = _wpack(w, groups)
      rowptr, entries, _ = sphere_adjoint(pos, Kh, Kw, stride, gy.shape[2:])
      check(lib().mode_sphere_conv_bwd_data_adj(ptr(gy), ptr(w), ptr(gx), ptr(wp), ptr(rowptr), ptr(entries), B, Ci, H, W, Co, Kh,
                                                Kw, Ho, Wo, G, 0 if overwrite else 1, stream_of(gy)), 'mode_sphere_conv_bwd_data_adj')
    return gx
  if overwrite:
    gx.zero_()  # the scatter form adds with atomics
  with torch.cuda.device_of(gy), profiling.region(_tag2('sphere_conv_bwd_data', w, gx), nbytes, flops, gy.device):
    wp = _wpack(w, groups)
    check(lib().mode_sphere_conv_bwd_data(ptr(gy), ptr(pos), ptr(w), ptr(gx), ptr(wp), *dims, stream_of(gy)),
          'mode_sphere_conv_bwd_data')
  return gx


SPHERE_POLAR = True  # polar kernel for the tall-window tiles (else: pixel list + general kernel)
SPHERE_BWD_WEIGHT_SPLIT = True  # compact-window tiles of the weight gradient on the split-bf16 kernel (CONV_ARITH 'bf16x6' only)


def _bww_win_entry():
  return 'mode_sphere_conv_bwd_weight_win_split' if (CONV_ARITH == 'bf16x6' and SPHERE_BWD_WEIGHT_SPLIT) else 'mode_sphere_conv_bwd_weight_win'


def _bww_f16_maxima(gy, x, gy_src=None, x_src=None):
  """(max |gy|, max |x|) buffers for the fp16 arithmetic of the windowed weight gradient (mode_sphere_conv_bwd_weight_win_split_f16), or
  None when it does not apply.  gy_src / x_src: the tensors the given ones are plane-transposed copies of (same values, maybe tagged)."""
  if not (SPHERE_BWD_F16 and CONV_ARITH == 'bf16x6' and SPHERE_BWD_WEIGHT_SPLIT):
    return None
  def of(t, src):
    am = known_abs_max(t)
    if am is None and src is not None and src is not t:
      am = known_abs_max(src)
    return am if am is not None else _tagged_abs_max(t)
  return (of(gy, gy_src), of(x, x_src))
SPHERE_BWD_WEIGHT = 'window'  # 'window' (where the table allows) | 'gather'


def sphere_conv_bwd_weight(gy, pos, x, gw, stride, groups, x_transposed=None, gy_transposed=None):
  """Accumulates into `gw` (Co,Ci/g,Kh,Kw) (caller zero-fills, sphere_conv.py:63).  x_transposed: the plane-transposed copy
  of x kept from the forward, if any (saves rebuilding it for the windowed kernel); gy_transposed: the same for gy."""
  require_gpu(gy, pos, x, gw)
  require_f32c(gy, pos, x, gw)
  dims = _sc_dims(x.shape, gw.shape, gy.shape[2:], stride, groups)
  B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, G = dims
  flops = 2 * gy.numel() * gw[0].numel()
  nbytes = 4 * (x.numel() + gy.numel() + pos.numel() + gw.numel())
  plan = None
  if SPHERE_BWD_WEIGHT == 'window' and (sH, sW) == (1, 1) and Kh * Kw == 9 and (Ho, Wo) == (H, W):
    plan = sphere_plan(pos, Kh, Kw)
    if plan is not None and plan[1][0] == 0:
      plan = None  # no compact tile at all: nothing to gain
    if not _plan_usable(plan):  # the same predicate as the forward and the input gradient: one layout per table
      post = sphere_native_t(pos, Kh, Kw)
      if post is not None:  # ERP-like table
        return sphere_conv_bwd_weight_t(gy, post, x, gw, groups)
  with torch.cuda.device_of(gy), profiling.region(_tag2('sphere_conv_bwd_weight', gw, x), nbytes, flops, gy.device):
    if plan is not None:
      tiles, (n0, n1, n2), rest, nrest, rec_w, rec_off, polar = plan
      if polar is not None and SPHERE_POLAR:  # the tall-window tiles on the polar kernel instead of the pixel list
        pitems, prw, pro, npol = polar
        nrest = 0
      else:
        pitems = prw = pro = None
        npol = 0
      n = lib().mode_sphere_conv_bwd_weight_win_workspace_bytes(B, Ci, H, W, Co, Kh, Kw, G, n0, nrest, npol)
      ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=gy.device)
      gyt = xt = None
      if SPHERE_LAYOUT == 'transposed':
        gyt = gy_transposed if gy_transposed is not None and tuple(gy_transposed.shape) == (B, Co, Wo, Ho) else transpose_planes(gy)
        xt = x_transposed if x_transposed is not None and tuple(x_transposed.shape) == (B, Ci, W, H) else transpose_planes(x)
      entry = _bww_win_entry()
      amax = _bww_f16_maxima(gyt if gyt is not None else gy, xt if xt is not None else x, gy, x)
      if amax is not None:
        entry, amax = entry + '_f16', (ptr(amax[0]), ptr(amax[1]))
      check(getattr(lib(), entry)(ptr(gy), ptr(pos), ptr(x), *(amax or ()), ptr(gw), ptr(ws), ptr(tiles), n0, n1, n2, ptr(rec_w),
                                  ptr(rec_off), ptr(rest), nrest, ptr(pitems) if npol else None, ptr(prw) if npol else None,
                                  ptr(pro) if npol else None, npol, B, Ci, H, W, Co, Kh, Kw, G,
                                  ptr(gyt) if gyt is not None else None, ptr(xt) if xt is not None else None,
                                  stream_of(gy)), entry)
    else:
      n = lib().mode_sphere_conv_bwd_weight_workspace_bytes(B, Ci, Co, Kh, Kw, Ho, Wo, G)
      ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=gy.device)
      check(lib().mode_sphere_conv_bwd_weight(ptr(gy), ptr(pos), ptr(x), ptr(gw), ptr(ws), *dims, stream_of(gy)),
            'mode_sphere_conv_bwd_weight')
  return gw


# ------------------------------------------------------------------------------------ plane-transposed operator
# A run of stride-1 3x3 spherical layers (layer4 of the extractor: 16 of them, with BatchNorm / ReLU / residual adds in
# between, all of which are indifferent to the order of the two spatial axes) can stay in the (B, C, W, H) storage the
# windowed kernels work on: the three functions below are the operator on that storage, without any transpose.
def sphere_t_supported(pos, w, B, groups):
  """True when the windowed forward would run for this table / weight / batch (else callers use the NCHW operator)."""
  if SPHERE_LAYOUT != 'transposed' or SPHERE_FWD != 'window' or SPHERE_BWD_WEIGHT != 'window' or SPHERE_BWD_DATA != 'gather':
    return False
  if w.shape[2] * w.shape[3] != 9:
    return False
  plan = sphere_plan(pos, w.shape[2], w.shape[3])
  if not _plan_usable(plan):
    return False
  return sum(plan[1]) * B * groups * (-(-(w.shape[0] // groups) // 128)) >= SPHERE_FWD_MIN_WG


def sphere_conv_fwd_t(xt, pos, w, yt, groups, f16=False, amax_out=None):
  """yt (B,Co,W,H) = spherical convolution of xt (B,Ci,W,H), both plane-transposed; stride 1, 3x3 taps.
  amax_out: a list that receives the (max |x|, max |w|) buffers when the fp16 arithmetic ran (for the layer's backward)."""
  require_gpu(xt, pos, w, yt)
  require_f32c(xt, pos, w, yt)
  B, Ci, W, H = xt.shape
  Co, _, Kh, Kw = w.shape
  assert tuple(pos.shape[2:]) == (H, W) and tuple(yt.shape) == (B, Co, W, H)
  tiles, (n0, n1, n2) = sphere_plan(pos, Kh, Kw)[:2]
  flops = 2 * yt.numel() * w[0].numel()
  nbytes = 4 * (xt.numel() + yt.numel() + pos.numel() + w.numel())
  with torch.cuda.device_of(xt), profiling.region('sphere_conv_fwd[%d->%d %dx%d]' % (Ci, Co, H, W), nbytes, flops, xt.device):
    wp = torch.empty(lib().mode_sphere_conv_win_wpack_bytes(Ci, Co, Kh, Kw, groups) // 4, dtype=torch.float32, device=w.device)
    amax = _sphere_f16_maxima(xt, w, f16)
    if amax is not None and amax_out is not None:
      amax_out.append(amax)
    _sphere_fwd_win(ptr(xt), pos, w, None, ptr(yt), wp, tiles, n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups, 1, stream_of(xt), amax)
  return yt


_adjplan_cache = _LRU(TABLE_CACHE_ENTRIES)
SPHERE_BWD_DATA_SPLIT = True  # windowed adjoint on the split-bf16 kernel where the plan allows (CONV_ARITH 'bf16x6' only)
SPHERE_BWD_SPLIT_MIN_WG = 200  # fewer 64 x 4 tiles x samples than this: the 64-pixel tiles of the gather kernel fill the chip better


def sphere_adjplan(pos, kh, kw):
  """Adjoint-window plan of the sampling table `pos` (H, W) for the windowed input-gradient kernel, or None when no tile qualifies
  (or H is not a multiple of 64: the left-over tiles are handed to the gather kernel as whole 64-pixel runs of one stored row).
  (good tiles int32 device, n_good, rec_off int32 device, rec_w float32 device, bad 64-pixel tile ids of the plane-transposed
  storage int32 device, n_bad_ids).  Built on the host by mode_sphere_adjplan_build once per (table, device), cached."""
  key = (pos.data_ptr(), pos._version, tuple(pos.shape), kh, kw, str(pos.device))
  with _plan_lock:
    if key in _adjplan_cache:
      return _adjplan_cache[key][0]
    H, W = pos.shape[2:]
    plan = None
    if kh * kw == 9 and H % 64 == 0 and W % 4 == 0:
      host = pos.detach().to('cpu', torch.float32).contiguous()
      n = lib().mode_sphere_plan_max_tiles(H, W)
      good = torch.zeros(4 * n, dtype=torch.int32)
      bad = torch.zeros(2 * n, dtype=torch.int32)
      counts = torch.zeros(2, dtype=torch.int32)
      rec_off = torch.zeros(n * 9 * 256 * 4, dtype=torch.int32)
      rec_w = torch.zeros(n * 9 * 256 * 4, dtype=torch.float32)
      rec_off2 = torch.zeros(n * 9 * 256 * 2, dtype=torch.int32)
      rec_w2 = torch.zeros(n * 9 * 256 * 2, dtype=torch.float32)
      check(lib().mode_sphere_adjplan_build(ptr(host), H, W, kh, kw, ptr(good), ptr(bad), ptr(counts), ptr(rec_off), ptr(rec_w), ptr(rec_off2),
                                            ptr(rec_w2)), 'mode_sphere_adjplan_build')
      ng, nb = int(counts[0]), int(counts[1])
      if ng > 0:
        # a bad tile (h0, w0) = the 64-pixel runs h0 .. h0 + 63 of the stored rows w0 .. w0 + 3 of the (W, H) planes
        b2 = bad[:2 * nb].view(nb, 2).to(torch.int64)
        ids = ((b2[:, 1:2] + torch.arange(4).view(1, 4)) * H + b2[:, 0:1]) // 64 if nb else torch.zeros((0, 4), dtype=torch.int64)
        ids = ids.reshape(-1).sort().values.to(torch.int32)
        dev = pos.device
        plan = (good[:4 * ng].contiguous().to(dev), ng, rec_off[:ng * 9 * 256 * 4].contiguous().to(dev),
                rec_w[:ng * 9 * 256 * 4].contiguous().to(dev), ids.contiguous().to(dev) if nb else None, int(ids.numel()),
                rec_off2[:ng * 9 * 256 * 2].contiguous().to(dev), rec_w2[:ng * 9 * 256 * 2].contiguous().to(dev))
    _adjplan_cache[key] = (plan, pos)  # keep `pos` alive: the key uses its address
    return plan


def sphere_conv_bwd_data_t(gyt, pos, w, gxt, groups, w_amax=None):
  """gxt (B,Ci,W,H) = input gradient for gyt (B,Co,W,H), both plane-transposed (written, not added to)."""
  require_gpu(gyt, pos, w, gxt)
  require_f32c(gyt, pos, w, gxt)
  B, Co, W, H = gyt.shape
  Ci, Kh, Kw = gxt.shape[1], w.shape[2], w.shape[3]
  flops = 2 * gyt.numel() * w[0].numel()
  nbytes = 4 * (gxt.numel() + gyt.numel() + pos.numel() + w.numel())
  rowptr, entries, _ = sphere_adjoint(_transposed_table(pos), Kh, Kw, (1, 1), (W, H))
  aplan = None
  if (CONV_ARITH == 'bf16x6' and SPHERE_BWD_DATA_SPLIT and Kh * Kw == 9 and tuple(pos.shape[2:]) == (H, W) and
      lib().mode_sphere_conv_bwd_data_win_supported(Ci, Co, groups) == 1):
    aplan = sphere_adjplan(pos, Kh, Kw)
    if aplan is not None and aplan[1] * B * groups * (-(-(Ci // groups) // 128)) < SPHERE_BWD_SPLIT_MIN_WG:
      aplan = None
  with torch.cuda.device_of(gyt), profiling.region('sphere_conv_bwd_data[%d->%d %dx%d]' % (Ci, Co, H, W), nbytes, flops, gyt.device):
    wp = _wpack(w, groups)
    if aplan is not None:
      tiles, ng, rec_off, rec_w, bad_ids, nbad, rec_off2, rec_w2 = aplan
      wps = torch.empty(lib().mode_sphere_conv_bwd_data_win_wpack_bytes(Ci, Co, Kh, Kw, groups) // 4, dtype=torch.float32, device=w.device)
      if SPHERE_BWD_F16:  # (a backward pass is a training step: the two-piece fp16 arithmetic)
        check(lib().mode_sphere_conv_bwd_data_win_split_f16(ptr(gyt), ptr(w), ptr(_tagged_abs_max(gyt)), ptr(_weight_abs_max(w, w_amax)), ptr(gxt),
                                                            ptr(wps), ptr(tiles), ng, ptr(rec_off), ptr(rec_w), ptr(rec_off2), ptr(rec_w2), B, Ci,
                                                            H, W, Co, Kh, Kw, groups, 1, stream_of(gyt)), 'mode_sphere_conv_bwd_data_win_split_f16')
      else:
        check(lib().mode_sphere_conv_bwd_data_win_split(ptr(gyt), ptr(w), ptr(gxt), ptr(wps), ptr(tiles), ng, ptr(rec_off), ptr(rec_w),
                                                        ptr(rec_off2), ptr(rec_w2), B, Ci, H, W, Co, Kh, Kw, groups, 1, stream_of(gyt)),
              'mode_sphere_conv_bwd_data_win_split')
      if nbad:  # the tiles next to the poles and the few columns with more than four sources per tap: gather kernel on a tile list
        check(lib().mode_sphere_conv_bwd_data_adj_list(ptr(gyt), ptr(w), ptr(gxt), ptr(wp), ptr(rowptr), ptr(entries), B, Ci, W, H, Co,
                                                       Kh, Kw, W, H, groups, 0, ptr(bad_ids), nbad, stream_of(gyt)),
              'mode_sphere_conv_bwd_data_adj_list')
    else:
      check(lib().mode_sphere_conv_bwd_data_adj(ptr(gyt), ptr(w), ptr(gxt), ptr(wp), ptr(rowptr), ptr(entries), B, Ci, W, H, Co, Kh, Kw,
                                                W, H, groups, 0, stream_of(gyt)), 'mode_sphere_conv_bwd_data_adj')
  return gxt


def sphere_conv_bwd_weight_t(gyt, pos, xt, gw, groups):
  """Adds the weight gradient for plane-transposed gyt (B,Co,W,H) and xt (B,Ci,W,H) to gw."""
  require_gpu(gyt, pos, xt, gw)
  require_f32c(gyt, pos, xt, gw)
  B, Co, W, H = gyt.shape
  Ci, Kh, Kw = xt.shape[1], gw.shape[2], gw.shape[3]
  tiles, (n0, n1, n2), rest, nrest, rec_w, rec_off, polar = sphere_plan(pos, Kh, Kw)
  if nrest and polar is None:
    raise RuntimeError('sphere_conv_bwd_weight_t: the table needs the pixel-list fallback, which reads NCHW tensors')
  pitems, prw, pro, npol = polar if polar is not None else (None, None, None, 0)
  flops = 2 * gyt.numel() * gw[0].numel()
  nbytes = 4 * (xt.numel() + gyt.numel() + pos.numel() + gw.numel())
  with torch.cuda.device_of(gyt), profiling.region('sphere_conv_bwd_weight[%d->%d %dx%d]' % (Ci, Co, H, W), nbytes, flops, gyt.device):
    n = lib().mode_sphere_conv_bwd_weight_win_workspace_bytes(B, Ci, H, W, Co, Kh, Kw, groups, n0, 0, npol)
    ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=gyt.device)
    entry = _bww_win_entry()
    amax = _bww_f16_maxima(gyt, xt)
    if amax is not None:
      entry, amax = entry + '_f16', (ptr(amax[0]), ptr(amax[1]))
    check(getattr(lib(), entry)(None, ptr(pos), None, *(amax or ()), ptr(gw), ptr(ws), ptr(tiles), n0, n1, n2, ptr(rec_w), ptr(rec_off),
                                ptr(rest), 0, ptr(pitems) if npol else None, ptr(prw) if npol else None,
                                ptr(pro) if npol else None, npol, B, Ci, H, W, Co, Kh, Kw, groups, ptr(gyt), ptr(xt),
                                stream_of(gyt)), entry)
  return gw


# ------------------------------------------------------------------------------------ the stride-2 3x3 layer (layer2[0].conv1)
def conv2d_3x3_s2_supported(x, conv):
  """Conv2d(k3, stride 2, padding 1) on even-sized planes, within the limits of the stride-1 kernels its gradients run on."""
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and
          conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros' and
          x.shape[2] % 2 == 0 and x.shape[3] % 4 == 0 and _conv2d_own(x, conv.weight) and conv2d_wgrad_supported(x, conv.weight))


class Conv2d3x3S2Function(torch.autograd.Function):
  """Forward on the integer-table gather kernel; both gradients as stride-1 gradients of the zero-inserted output gradient, on the
  MFMA kernels of the stride-1 layers (csrc/conv2d.hip, conv2d_wgrad.hip)."""

  @staticmethod
  def forward(ctx, x, w):
    x, w = x.contiguous(), w.contiguous()
    B, Ci, H, W = x.shape
    y = torch.empty((B, w.shape[0], H // 2, W // 2), dtype=x.dtype, device=x.device)
    sphere_conv_fwd(x, conv2d_table(H, W, 3, 3, (2, 2), (1, 1), (1, 1), x.device), w, y, (2, 2), 1)
    ctx.save_for_backward(x, w)
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    gy = gy.contiguous()
    B, Co, Ho, Wo = gy.shape
    up = torch.empty((B, Co, 2 * Ho, 2 * Wo), dtype=gy.dtype, device=gy.device)
    with torch.cuda.device_of(gy):
      check(lib().mode_zero_insert2(ptr(gy), ptr(up), B * Co, Ho, Wo, stream_of(gy)), 'mode_zero_insert2')
    gx = conv2d_bwd_data(up, w, 1) if ctx.needs_input_grad[0] else None
    gw = None
    if ctx.needs_input_grad[1]:
      sink = grad_sink(w)
      gw = conv2d_bwd_weight(up, x, 1, into=sink)
      if sink is not None:
        gw = None
    return gx, gw


def conv2d_3x3_s2(x, conv):
  return Conv2d3x3S2Function.apply(x, conv.weight)


# ------------------------------------------------------------------------------------ 1x1 Conv2d (stride 1 | 2): plain MFMA GEMMs
def conv1x1_supported(x, conv):
  """The extractor's 1x1 layers (downsample branches, lastconv[0], lastconv[4]) on csrc/conv1x1.hip; the weight gradient loads
  16-byte row pieces, hence the divisibility conditions (other shapes run on the integer-table kernels)."""
  if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (1, 1) and conv.groups == 1 and
          conv.bias is None and conv.padding == (0, 0) and conv.stride in ((1, 1), (2, 2))):
    return False
  H, W = x.shape[2:]
  s = conv.stride[0]
  return ((W - 1) // s + 1) % 4 == 0 and (s == 1 or W % 8 == 0) and max(conv.in_channels, conv.out_channels) * H * W < 2**31


def _tag1(name, ci, co, s, h, w):
  return '%s[%d->%d s%d %dx%d]' % (name, ci, co, s, h, w) if profiling.ENABLED else name


def conv1x1_fwd(x, w, stride=1, bn=None, add=None, relu=False):
  """y = conv2d(x, w (Co, Ci, 1, 1), stride); with bn: relu?(eval-mode bn(y) [+ add]) in the same launch."""
  require_gpu(x, w)
  x, w = x.contiguous(), w.detach().contiguous() if bn is not None else w.contiguous()
  require_f32c(x, w)
  B, Ci, H, W = x.shape
  Co = w.shape[0]
  y = torch.empty((B, Co, (H - 1) // stride + 1, (W - 1) // stride + 1), dtype=x.dtype, device=x.device)
  flops = 2 * y.numel() * Ci
  with torch.cuda.device_of(x), profiling.region(_tag1('conv1x1_fwd', Ci, Co, stride, H, W), 4 * (x.numel() // stride**2 + y.numel() + w.numel()),
                                                 flops, x.device):
    if bn is None:
      wp = torch.empty(lib().mode_conv1x1_wpack_bytes(Ci, Co) // 4, dtype=torch.float32, device=x.device)
      check(lib().mode_conv1x1_fwd(ptr(x), ptr(w), ptr(y), ptr(wp), B, Ci, H, W, Co, stride, stream_of(x)), 'mode_conv1x1_fwd')
    else:
      e, keep = _epilogue(bn, add, relu, y)
      wp, reuse = _eval_wpack(bn, 'conv1x1_fwd_bn', w, lib().mode_conv1x1_wpack_bytes(Ci, Co) // 4, x.device)
      with reuse:
        check(lib().mode_conv1x1_fwd_bn(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, H, W, Co, stride, stream_of(x)),
              'mode_conv1x1_fwd_bn')
  return y


class Conv1x1Function(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, w, stride):
    ctx.save_for_backward(x, w)
    ctx.stride = stride
    return conv1x1_fwd(x, w, stride)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    x, w, gy = x.contiguous(), w.contiguous(), gy.contiguous()
    B, Ci, H, W = x.shape
    Co, s = w.shape[0], ctx.stride
    gx = None
    if ctx.needs_input_grad[0]:
      gx = torch.empty_like(x)
      with torch.cuda.device_of(x), profiling.region(_tag1('conv1x1_bwd_data', Ci, Co, s, H, W), 4 * (gx.numel() + gy.numel() + w.numel()),
                                                     2 * gy.numel() * Ci, x.device):
        wp = torch.empty(lib().mode_conv1x1_wpack_bytes(Ci, Co) // 4, dtype=torch.float32, device=x.device)
        check(lib().mode_conv1x1_bwd_data(ptr(gy), ptr(w), ptr(gx), ptr(wp), B, Ci, H, W, Co, s, stream_of(x)), 'mode_conv1x1_bwd_data')
    gw = None
    if ctx.needs_input_grad[1]:
      sink = grad_sink(w)
      gw = sink if sink is not None else torch.empty_like(w)
      with torch.cuda.device_of(x), profiling.region(_tag1('conv1x1_bwd_weight', Ci, Co, s, H, W), 4 * (x.numel() // s**2 + gy.numel() + w.numel()),
                                                     2 * gy.numel() * Ci, x.device):
        ws = torch.empty(max(lib().mode_conv1x1_bwd_weight_workspace_bytes(B, Ci, H, W, Co, s) // 4, 1), dtype=torch.float32, device=x.device)
        check(lib().mode_conv1x1_bwd_weight(ptr(gy), ptr(x), ptr(gw), ptr(ws), B, Ci, H, W, Co, s, int(sink is not None), stream_of(x)),
              'mode_conv1x1_bwd_weight')
      if sink is not None:
        gw = None
    return gx, gw, None


def conv1x1(x, conv):
  return Conv1x1Function.apply(x, conv.weight, conv.stride[0])


# ------------------------------------------------------------------------------------ the 7x7 stride-2 stem (3 -> 32)
def conv_stem_supported(x, conv):
  """firstconv[0] (submodule.py:155): Conv2d(3, <= 32, 7, stride 2, padding 3), on an input that needs no gradient."""
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and
          conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.in_channels == 3 and
          conv.out_channels <= 32 and conv.padding_mode == 'zeros' and not (torch.is_grad_enabled() and x.requires_grad) and
          32 * x.shape[2] * x.shape[3] < 2**31)


def conv_stem_fwd(x, w, bn=None, add=None, relu=False):
  require_gpu(x, w)
  x, w = x.contiguous(), w.detach().contiguous() if bn is not None else w.contiguous()
  require_f32c(x, w)
  B, Ci, H, W = x.shape
  Co = w.shape[0]
  y = torch.empty((B, Co, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device)
  with torch.cuda.device_of(x), profiling.region(_tag1('conv_stem_fwd', Ci, Co, 2, H, W), 4 * (x.numel() + y.numel() + w.numel()),
                                                 2 * y.numel() * Ci * 49, x.device):
    if bn is None:
      wp = torch.empty(lib().mode_conv_stem_wpack_bytes(Ci, Co) // 4, dtype=torch.float32, device=x.device)
      check(lib().mode_conv_stem_fwd(ptr(x), ptr(w), ptr(y), ptr(wp), B, Ci, H, W, Co, stream_of(x)), 'mode_conv_stem_fwd')
    else:
      e, keep = _epilogue(bn, add, relu, y)
      wp, reuse = _eval_wpack(bn, 'conv_stem_fwd_bn', w, lib().mode_conv_stem_wpack_bytes(Ci, Co) // 4, x.device)
      with reuse:
        check(lib().mode_conv_stem_fwd_bn(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, H, W, Co, stream_of(x)),
              'mode_conv_stem_fwd_bn')
  return y


class ConvStemFunction(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, w):
    ctx.save_for_backward(x, w)
    return conv_stem_fwd(x, w)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    if ctx.needs_input_grad[0]:
      raise RuntimeError('conv_stem: the stem kernels have no input gradient (conv_stem_supported excludes such inputs)')
    gw = None
    if ctx.needs_input_grad[1]:
      x, gy = x.contiguous(), gy.contiguous()
      B, Ci, H, W = x.shape
      Co = w.shape[0]
      sink = grad_sink(w)
      gw = sink if sink is not None else torch.empty_like(w)
      with torch.cuda.device_of(x), profiling.region(_tag1('conv_stem_bwd_weight', Ci, Co, 2, H, W), 4 * (x.numel() + gy.numel() + w.numel()),
                                                     2 * gy.numel() * Ci * 49, x.device):
        ws = torch.empty(max(lib().mode_conv_stem_bwd_weight_workspace_bytes(B, Ci, H, W, Co) // 4, 1), dtype=torch.float32, device=x.device)
        check(lib().mode_conv_stem_bwd_weight(ptr(gy), ptr(x), ptr(gw), ptr(ws), B, Ci, H, W, Co, int(sink is not None), stream_of(x)),
              'mode_conv_stem_bwd_weight')
      if sink is not None:
        gw = None
    return None, gw


def conv_stem(x, conv):
  return ConvStemFunction.apply(x, conv.weight)


# ------------------------------------------------------------------------------------ any other Conv2d: gather-and-MAC on an integer table
# The extractor's remaining regular convolutions -- 7x7 stride 2 (stem), 3x3 stride 2, 1x1 (stride 1 and 2); submodule.py:155,
# 158, 162, 167-174 -- are the spherical operator with an INTEGER sampling table: tap (i, j) of output pixel (h, w) reads input
# pixel (h*s + i*d - p, w*s + j*d - p), the bilinear weights are exactly (1, 0, 0, 0) and positions outside the image are dropped
# by the operator's own guard (cu:246), which is zero padding.  Same kernels as SphereConv (general gather-and-MAC forward,
# adjoint-gather input gradient, split-K weight gradient), bit-for-bit a plain convolution up to the order of the fp32 sums.
_conv_tables = _LRU(TABLE_CACHE_ENTRIES)
_conv_table_lock = threading.Lock()


MAX_TAPS = 32  # tap limit of the general gather-and-MAC kernels (csrc/sphere_conv.hip)


def conv2d_table(H, W, kh, kw, stride, pad, dil, device, row0=0):
  """(1, 2*kh*kw, H, W) float32 table on `device`: channel 2k = row, 2k+1 = column read by tap k = (i, j) at the output pixel
  whose top-left input position is (h, w) (the operator samples it at (h_out*stride, w_out*stride), cu:206-261); tap rows are
  numbered from `row0` (a kernel with more than MAX_TAPS taps runs as several row bands).  Cached."""
  key = (H, W, kh, kw, tuple(stride), tuple(pad), tuple(dil), str(device), row0)
  with _conv_table_lock:
    t = _conv_tables.get(key)
    if t is None:
      hh = torch.arange(H, dtype=torch.float32).view(H, 1).expand(H, W)
      ww = torch.arange(W, dtype=torch.float32).view(1, W).expand(H, W)
      planes = []
      for i in range(row0, row0 + kh):
        for j in range(kw):
          planes += [hh + float(i * dil[0] - pad[0]), ww + float(j * dil[1] - pad[1])]
      t = _conv_tables[key] = torch.stack(planes, 0).unsqueeze(0).contiguous().to(device)
    return t


def conv2d_tabled_supported(x, conv):
  """The operator reads its table at (h_out * stride, w_out * stride): the output grid must fit into the input grid (any 'same'
  or shrinking geometry; a convolution padded beyond that stays with the vendor library)."""
  if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.groups == 1 and conv.bias is None and
          conv.padding_mode == 'zeros' and not isinstance(conv.padding, str) and conv.kernel_size[1] <= MAX_TAPS and
          x.shape[2] >= conv.kernel_size[0] and x.shape[3] >= conv.kernel_size[1]):
    return False
  for n, k, s, p, d in zip(x.shape[2:], conv.kernel_size, conv.stride, conv.padding, conv.dilation):
    out = (n + 2 * p - (d * (k - 1) + 1)) // s + 1
    if out < 1 or (out - 1) * s > n - 1:
      return False
  return True


def _row_bands(kh, kw):
  """[(first row, rows)] with rows * kw <= MAX_TAPS each: 3x3 -> [(0, 3)], 7x7 -> [(0, 4), (4, 3)]."""
  per = max(1, MAX_TAPS // kw)
  return [(r, min(per, kh - r)) for r in range(0, kh, per)]


class Conv2dTabledFunction(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, w, stride, pad, dil):
    x, w = x.contiguous(), w.contiguous()
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    y = torch.empty((B, Co, Ho, Wo), dtype=x.dtype, device=x.device)
    bands = _row_bands(kh, kw)
    for r0, rows in bands:
      pos = conv2d_table(H, W, rows, kw, stride, pad, dil, x.device, r0)
      if len(bands) == 1:
        sphere_conv_fwd(x, pos, w, y, stride, 1)
      else:  # y = sum over the row bands of the kernel
        part = y if r0 == 0 else torch.empty_like(y)
        sphere_conv_fwd(x, pos, w[:, :, r0:r0 + rows].contiguous(), part, stride, 1)
        if r0:
          y += part
    ctx.save_for_backward(x, w)
    ctx.geom = (tuple(stride), tuple(pad), tuple(dil))
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    stride, pad, dil = ctx.geom
    H, W = x.shape[2:]
    kh, kw = w.shape[2:]
    gy = gy.contiguous()
    bands = _row_bands(kh, kw)
    gx = None
    if ctx.needs_input_grad[0]:
      gx = torch.empty_like(x)
      for r0, rows in bands:
        pos = conv2d_table(H, W, rows, kw, stride, pad, dil, x.device, r0)
        if len(bands) == 1:
          sphere_conv_bwd_data(gy, pos, w, gx, stride, 1, overwrite=True)
        else:
          part = gx if r0 == 0 else torch.empty_like(gx)
          sphere_conv_bwd_data(gy, pos, w[:, :, r0:r0 + rows].contiguous(), part, stride, 1, overwrite=True)
          if r0:
            gx += part
    gw = None
    if ctx.needs_input_grad[1]:
      sink = grad_sink(w)
      gw = sink if sink is not None else torch.zeros_like(w)
      for r0, rows in bands:
        pos = conv2d_table(H, W, rows, kw, stride, pad, dil, x.device, r0)
        if len(bands) == 1:
          sphere_conv_bwd_weight(gy, pos, x, gw, stride, 1)
        else:
          part = torch.zeros((w.shape[0], w.shape[1], rows, kw), dtype=w.dtype, device=w.device)
          sphere_conv_bwd_weight(gy, pos, x, part, stride, 1)
          gw[:, :, r0:r0 + rows] += part
      if sink is not None:
        gw = None
    return gx, gw, None, None, None


def conv2d_tabled(x, conv):
  """nn.Conv2d `conv` (any kernel size / stride / padding / dilation; groups 1, no bias) on the gather-and-MAC kernels."""
  return Conv2dTabledFunction.apply(x, conv.weight, tuple(conv.stride), tuple(conv.padding), tuple(conv.dilation))


# ------------------------------------------------------------------------------------ gradient sinks
def grad_sink(param):
  """The buffer parameter gradients are accumulated into directly by the native backward kernels, or None.

  autograd's AccumulateGrad costs one extra elementwise kernel per parameter and step (243 launches for ModeDisparity).
  When a parameter carries `_mode_grad_sink` (set by data_parallel.GradAllReducer: a view into its flat gradient buffer,
  which is also `param.grad`), the weight-gradient kernels add into that view themselves and the autograd Function
  returns None for the parameter, so nothing is left for autograd to accumulate.  Parameters without the attribute get
  ordinary autograd gradients."""
  sink = getattr(param, '_mode_grad_sink', None)
  if sink is None:
    return None
  if param.grad is None or param.grad.data_ptr() != sink.data_ptr():
    # optimizer.zero_grad() (set_to_none=True by default; train_disparity.py:149) or anything else replaced .grad: adding into
    # the detached buffer would silently drop this parameter's gradient, so autograd accumulates it the ordinary way
    # (GradAllReducer.rebind() re-attaches the views)
    return None
  if not (sink.is_cuda and sink.dtype == torch.float32 and sink.is_contiguous() and sink.shape == param.shape):
    raise RuntimeError('gradient sink of a %s parameter must be a contiguous fp32 device tensor of the same shape' % (tuple(param.shape),))
  return sink


# ------------------------------------------------------------------------------------ tensors with several consumers
def sum_n(tensors):
  """a + b [+ c [+ d]] in one pass (mode_sum_n); more than four operands in groups of four."""
  ts = [t.contiguous() for t in tensors]
  require_gpu(*ts)
  require_f32c(*ts)
  while len(ts) > 1:
    grp, ts = ts[:4], ts[4:]
    out = torch.empty_like(grp[0])
    with torch.cuda.device_of(out), profiling.region('grad_sum%d' % len(grp), 4 * out.numel() * (len(grp) + 1), 0, out.device):
      check(lib().mode_sum_n(ptr(grp[0]), ptr(grp[1]), ptr(grp[2]) if len(grp) > 2 else None, ptr(grp[3]) if len(grp) > 3 else None, ptr(out),
                             out.numel(), stream_of(out)), 'mode_sum_n')
    ts = [out] + ts
  return ts[0]


class FanOutFunction(torch.autograd.Function):
  """n aliases of x, one per consumer.  autograd would accumulate the consumers' gradients pairwise (n - 1 launches, each reading two
  tensors and writing one); here they arrive together and are summed in ONE pass.  Pays from three consumers on: cost0 (the input of
  dres2 and three residual adds) and pre1 (mode_disparity.py:119-125); exact up to the association of the fp32 sum."""

  @staticmethod
  def forward(ctx, x, n):
    return tuple(x.view_as(x) for _ in range(n))

  @staticmethod
  def backward(ctx, *grads):
    live = [g for g in grads if g is not None]
    if not live:
      return None, None
    return (live[0] if len(live) == 1 else sum_n(live)), None


# A tensor with exactly TWO consumers whose gradients autograd would add in a separate pass (read two tensors, write one): the consumer
# whose backward runs first leaves its gradient in the carrier and reports None, the second adds it inside the kernel that produces its
# own (mode_conv3d_bwd_data_split_acc: one extra read in the store instead of three passes) -- the same fp32 sum, bit for bit.  Both
# consumers register in their forward (arm); a gradient is only left behind when both did.  Consumers that cannot add in their kernel
# (the BatchNorm backward that owns a residual skip's gradient) can still be the first.
# The hand-off relies on both backwards running in the same pass -- what this model's training step guarantees -- and CHECKS it (ADVICE
# r5): the consumer that leaves a gradient queues a callback for the end of that backward pass; a gradient still parked then means the
# partner never ran (backward(inputs=...), autograd.grad on a sub-graph): the pass has reported None for a gradient it dropped, and
# the callback raises instead of letting a wrong gradient through (and un-parks the tensor, up to 403 MB).  A gradient left over from a
# pass that died before its callbacks ran is recognised by its owner -- the same consumer asking first again -- and dropped.
GRAD_CARRIERS = True  # bench.py --no-grad-carriers measures autograd's own accumulation


class GradCarrier(object):
  __slots__ = ('armed', 'grad', 'owner')

  def __init__(self):
    self.armed = 0
    self.grad = None
    self.owner = None

  def arm(self, needs_grad):
    if needs_grad:
      self.armed += 1

  def take(self, who=None):
    """The partner's gradient when it came first, else None.  `who` identifies the asking consumer (its autograd ctx): a gradient it
    parked itself is a leftover of an earlier, aborted pass -- dropped, not added."""
    g, owner = self.grad, self.owner
    self.grad = self.owner = None
    if g is not None and who is not None and owner is who:
      return None
    return g

  def leave(self, g, who=None):
    """First of the two backwards: True when g was left for the other one (report None to autograd), False to return it as usual."""
    if self.armed == 2 and g is not None:
      self.grad, self.owner = g, who
      try:
        torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
      except RuntimeError:  # (not inside a backward pass: nothing to check against; behave like autograd)
        self.grad = self.owner = None
        return False
      return True
    return False

  def _end_of_pass(self):
    if self.grad is not None:
      self.grad = self.owner = None
      raise RuntimeError('mode_hip GradCarrier: one of the two consumers of a shared tensor ran its backward and the other did not '
                         '(backward(inputs=...) / autograd.grad on a sub-graph?): the gradient of the shared tensor would be incomplete. '
                         'Set mode_hip.functional.GRAD_CARRIERS = False for partial backward passes.')


def grad_carrier(x):
  """A carrier for tensor x, or None when nothing would be gained (no gradient needed, CPU tensor, switched off)."""
  return GradCarrier() if (GRAD_CARRIERS and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad) else None


def fan_out(x, n):
  if n <= 1 or not (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad):
    return (x,) * max(n, 1)
  return FanOutFunction.apply(x, n)


# ------------------------------------------------------------------------------------ 3x3x3 convolution / transposed conv
def _wpack3d(ci, co, device):
  n = lib().mode_conv3d_wpack_bytes(ci, co)
  return torch.empty(n // 4, dtype=torch.float32, device=device)


# Arithmetic of the 3x3(x3) stride-1 convolution layers -- the 3D regulariser's stride-1 layers (forward, input gradient, weight
# gradient), the extractor's regular 3x3 Conv2d layers (all three) and the forward of its spherical layers (small-window tiles):
#   'bf16x6' (default) fp32 operands split EXACTLY into three bf16 pieces when a tile is staged, six bf16 MFMAs per product (the
#            terms >= 2^-16 of it), fp32 accumulation (csrc/conv3d_split.hip, conv3d_split_wgrad.hip, conv2d_split.hip).  Results carry
#            the rounding of an fp32 convolution -- measured against float64 they are at least as close as the fp32 MFMA kernels' on
#            the same inputs (tests/test_gpu_split.py) -- at 1.5-2 x their speed.  Layers the split kernels do not cover (stride 2,
#            transposed, single-channel heads, channel counts off their grid) run on the fp32 kernels.
#   'f32'    v_mfma_f32_32x32x2_f32 everywhere (csrc/conv3d.hip, conv2d.hip).
# Process-wide, read at call time (bench.py --conv-arith; the parity tests run under both).
CONV_ARITH = 'bf16x6'


def set_conv_arith(kind):
  global CONV_ARITH
  if kind not in ('f32', 'bf16x6'):
    raise ValueError("convolution arithmetic must be 'f32' or 'bf16x6', got %r" % (kind,))
  CONV_ARITH = kind


def _split3d(ci, co, stride, which):
  """which: 0 forward, 1 input gradient, 2 weight gradient."""
  return CONV_ARITH == 'bf16x6' and lib().mode_conv3d_split_supported(ci, co, stride, int(which)) == 1


def _deconv_split_fits(lowres_voxels, cout):
  """The 32-bit offset limits of deconv3d_split (csrc/conv3d_split_deconv.hip: MODE_REQUIRE in deconv3d_split), in the voxels of its
  LOW-resolution input and the channels of its output; volumes beyond them run on the fp32 kernel instead of failing."""
  return lowres_voxels * 8 * max(cout, 8) < 2**31 and lowres_voxels < 2**27


def _out3(n, stride):
  return (n - 1) // stride + 1


def _tag3(name, ci, co, stride, d, h, w):
  """Profiling label carrying the layer shape (input volume of the convolution), e.g. conv3d_fwd[32->32 s1 48x256x128]."""
  return '%s[%d->%d s%d %dx%dx%d]' % (name, ci, co, stride, d, h, w) if profiling.ENABLED else name


# The stride-1 3-D layers of a TRAINING step run on TWO fp16 pieces and three MFMAs per product (mode_conv3d_*_split_f16; the
# product default since round 5, `bench.py --no-conv3d-f16` is the A/B): each operand scaled by a power of two taken from its tensor's
# largest finite magnitude -- left by the BatchNorm pass that wrote the tensor (the `_amax` entries), or by mode_abs_max.
# Precision contract (include/mode_hip.h): an element keeps 22 significant bits down to ~2^-17 of its tensor's maximum, fewer below,
# none below ~2^-39 of it; three bf16 pieces (CONV3D_S1_F16 = False, and every inference call) keep 24 bits for every element.
CONV3D_S1_F16 = True
# ... and of an INFERENCE forward (conv3d_bn_eval: the folded-BatchNorm epilogues on the same arithmetic, the activations' maxima out of the
# kernels' epilogues; mode_conv3d_fwd_split_f16_bn)
CONV3D_EVAL_F16 = True
CONV2D_EVAL_F16 = True  # (the extractor's stride-1 3 x 3 layers of an inference forward: mode_conv2d_fwd_split_f16_bn)


BN_ABSMAX_FLOATS = 2064  # MODE_BN_ABSMAX_FLOATS of include/mode_hip.h: the buffer a tensor's maximum lives in


def abs_max(t):
  """The largest finite magnitude in t as the fp16-arithmetic kernels take it (mode_abs_max; no host synchronisation): a device buffer
  of BN_ABSMAX_FLOATS floats whose maximum is the value (abs_max_value reads it on the host).  A tensor's maximum is computed once and
  handed to every kernel that reads the tensor (the forward's x again in the weight gradient, gy in both gradients)."""
  out = torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=t.device)
  check(lib().mode_abs_max(ptr(t), t.numel(), ptr(out), stream_of(t)), 'mode_abs_max')
  return out


def abs_max_value(buf):
  """Host value of a maximum buffer (abs_max, known_abs_max): its entries are non-negative floats, the value is their maximum.  Synchronises."""
  return float(buf.max())


def conv3d_s1_f16(ci, co, which):
  """True when the stride-1 layer ci -> co runs its forward (which 0) / input gradient (1) / weight gradient (2) on the fp16 arithmetic."""
  return CONV3D_S1_F16 and _split3d(ci, co, 1, which)


def conv3d_fwd(x, w, stride=1, amax=None):
  """x (B,Ci,D,H,W), w (Co,Ci,3,3,3) -> (B,Co,Do,Ho,Wo); k3 p1, stride 1|2, no bias (convbn_3d, submodule.py:20-22).
  amax = (max |x|, max |w|) device scalars when the caller has them (fp16 arithmetic only; computed here otherwise)."""
  require_gpu(x, w)
  x, w = x.contiguous(), w.contiguous()
  require_f32c(x, w)
  B, Ci, D, H, W = x.shape
  Co = w.shape[0]
  if tuple(w.shape[1:]) != (Ci, 3, 3, 3):
    raise RuntimeError('conv3d: weight %s does not match input channels %d / kernel 3' % (tuple(w.shape), Ci))
  y = torch.empty((B, Co, _out3(D, stride), _out3(H, stride), _out3(W, stride)), dtype=x.dtype, device=x.device)
  flops = 2 * y.numel() * Ci * 27
  with torch.cuda.device_of(x), profiling.region(_tag3('conv3d_fwd', Ci, Co, stride, D, H, W), 4 * (x.numel() + y.numel() + w.numel()),
                                                 flops, x.device):
    wp = _wpack3d(Ci, Co, x.device)
    if _split3d(Ci, Co, stride, False) and stride == 2 and D * H * W < 2**26:
      check(lib().mode_conv3d_fwd_s2_split(ptr(x), ptr(w), None, ptr(y), ptr(wp), B, Ci, D, H, W, Co, stream_of(x)),
            'mode_conv3d_fwd_s2_split')
    elif _split3d(Ci, Co, stride, False) and stride == 1 and CONV3D_S1_F16:
      ax, aw = amax if amax is not None else (abs_max(x), abs_max(w))
      check(lib().mode_conv3d_fwd_split_f16(ptr(x), ptr(w), ptr(ax), ptr(aw), ptr(y), ptr(wp), B, Ci, D, H, W, Co, stream_of(x)),
            'mode_conv3d_fwd_split_f16')
    elif _split3d(Ci, Co, stride, False) and stride == 1:
      check(lib().mode_conv3d_fwd_split(ptr(x), ptr(w), None, ptr(y), ptr(wp), B, Ci, D, H, W, Co, stream_of(x)), 'mode_conv3d_fwd_split')
    else:
      check(lib().mode_conv3d_fwd(ptr(x), ptr(w), ptr(y), ptr(wp), B, Ci, D, H, W, Co, stride, stream_of(x)), 'mode_conv3d_fwd')
  return y


def conv3d_bwd_data(gy, w, in_shape, stride=1, acc=None, amax=None):
  """gradient w.r.t. the input of conv3d_fwd; in_shape = x.shape.  acc: a gradient of the same tensor that is already there -- the sum
  is returned (added in the kernel's store on the split path, by a separate pass elsewhere)."""
  require_gpu(gy, w)
  gy, w = gy.contiguous(), w.contiguous()
  require_f32c(gy, w)
  B, Ci, D, H, W = in_shape
  Co = w.shape[0]
  gx = torch.empty((B, Ci, D, H, W), dtype=gy.dtype, device=gy.device)
  flops = 2 * gy.numel() * Ci * 27
  if acc is not None:
    acc = acc.contiguous()
    s2_ok = stride == 2 and D % 2 == 0 and H % 2 == 0 and W % 2 == 0 and _deconv_split_fits(D * H * W // 8, Ci)
    if (tuple(acc.shape) == tuple(gx.shape) and acc.dtype == gx.dtype and CONV_ARITH == 'bf16x6' and (stride == 1 or s2_ok) and
        lib().mode_conv3d_bwd_data_split_acc_supported(Ci, Co, stride) == 1):
      with torch.cuda.device_of(gy), profiling.region(_tag3('conv3d_bwd_data', Ci, Co, stride, D, H, W),
                                                      4 * (2 * gx.numel() + gy.numel() + w.numel()), flops, gy.device):
        wp = _wpack3d(Ci, Co, gy.device)
        if stride == 1 and CONV3D_S1_F16:
          ag, aw = amax if amax is not None else (abs_max(gy), abs_max(w))
          check(lib().mode_conv3d_bwd_data_split_f16(ptr(gy), ptr(w), ptr(ag), ptr(aw), ptr(acc), ptr(gx), ptr(wp), B, Ci, D, H, W, Co,
                                                     stream_of(gy)), 'mode_conv3d_bwd_data_split_f16')
        else:
          check(lib().mode_conv3d_bwd_data_split_acc(ptr(gy), ptr(w), ptr(acc), ptr(gx), ptr(wp), B, Ci, D, H, W, Co, stride, stream_of(gy)),
                'mode_conv3d_bwd_data_split_acc')
      return gx
    return conv3d_bwd_data(gy, w, in_shape, stride, amax=amax).add_(acc)
  with torch.cuda.device_of(gy), profiling.region(_tag3('conv3d_bwd_data', Ci, Co, stride, D, H, W),
                                                  4 * (gx.numel() + gy.numel() + w.numel()), flops, gy.device):
    wp = _wpack3d(Ci, Co, gy.device)
    if stride == 2 and _split3d(Ci, Co, stride, True) and D % 2 == 0 and H % 2 == 0 and W % 2 == 0 and _deconv_split_fits(D * H * W // 8, Ci):
      check(lib().mode_conv3d_bwd_data_s2_split(ptr(gy), ptr(w), ptr(gx), ptr(wp), B, Ci, D, H, W, Co, stream_of(gy)),
            'mode_conv3d_bwd_data_s2_split')
    elif stride == 1 and _split3d(Ci, Co, stride, True) and CONV3D_S1_F16:
      ag, aw = amax if amax is not None else (abs_max(gy), abs_max(w))
      check(lib().mode_conv3d_bwd_data_split_f16(ptr(gy), ptr(w), ptr(ag), ptr(aw), None, ptr(gx), ptr(wp), B, Ci, D, H, W, Co, stream_of(gy)),
            'mode_conv3d_bwd_data_split_f16')
    elif stride == 1 and _split3d(Ci, Co, stride, True):
      check(lib().mode_conv3d_bwd_data_split(ptr(gy), ptr(w), ptr(gx), ptr(wp), B, Ci, D, H, W, Co, stream_of(gy)),
            'mode_conv3d_bwd_data_split')
    else:
      check(lib().mode_conv3d_bwd_data(ptr(gy), ptr(w), ptr(gx), ptr(wp), B, Ci, D, H, W, Co, stride, stream_of(gy)),
            'mode_conv3d_bwd_data')
  return gx


def conv3d_bwd_weight(gy, x, stride=1, into=None, amax=None):
  """gW (Co,Ci,3,3,3) = sum gy[o, q] * x[c, stride*q + k - 1]; returned, or ADDED to `into` when given.
  amax = (max |gy|, max |x|) device scalars when the caller has them (fp16 arithmetic only)."""
  require_gpu(gy, x)
  gy, x = gy.contiguous(), x.contiguous()
  require_f32c(gy, x)
  B, Ci, D, H, W = x.shape
  Co = gy.shape[1]
  gw = into if into is not None else torch.empty((Co, Ci, 3, 3, 3), dtype=gy.dtype, device=gy.device)
  if gw.numel() != Co * Ci * 27:
    raise RuntimeError('conv3d_bwd_weight: gradient buffer %s does not hold %dx%dx27 values' % (tuple(gw.shape), Co, Ci))
  flops = 2 * gy.numel() * Ci * 27
  with torch.cuda.device_of(gy), profiling.region(_tag3('conv3d_bwd_weight', Ci, Co, stride, D, H, W),
                                                  4 * (x.numel() + gy.numel() + gw.numel()), flops, gy.device):
    n = lib().mode_conv3d_bwd_weight_workspace_bytes(B, Ci, D, H, W, Co, stride)
    ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=gy.device)
    if stride == 2 and _split3d(Ci, Co, 2, 2) and D % 2 == 0 and H % 2 == 0 and W % 8 == 0 and 32 * D * H * W < 2**29:
      check(lib().mode_conv3d_bwd_weight_s2_split(ptr(gy), ptr(x), ptr(gw), ptr(ws), B, Ci, D, H, W, Co, int(into is not None), stream_of(gy)),
            'mode_conv3d_bwd_weight_s2_split')
    elif stride == 1 and _split3d(Ci, Co, stride, 2) and max(Ci, Co) * D * H * W < 2**29 and CONV3D_S1_F16:
      ag, ax = amax if amax is not None else (abs_max(gy), abs_max(x))
      check(lib().mode_conv3d_bwd_weight_split_f16(ptr(gy), ptr(x), ptr(ag), ptr(ax), ptr(gw), ptr(ws), B, Ci, D, H, W, Co, int(into is not None),
                                                   stream_of(gy)), 'mode_conv3d_bwd_weight_split_f16')
    elif stride == 1 and _split3d(Ci, Co, stride, 2) and max(Ci, Co) * D * H * W < 2**29:
      check(lib().mode_conv3d_bwd_weight_split(ptr(gy), ptr(x), ptr(gw), ptr(ws), B, Ci, D, H, W, Co, int(into is not None), stream_of(gy)),
            'mode_conv3d_bwd_weight_split')
    else:
      check(lib().mode_conv3d_bwd_weight(ptr(gy), ptr(x), ptr(gw), ptr(ws), B, Ci, D, H, W, Co, stride, int(into is not None), stream_of(gy)),
            'mode_conv3d_bwd_weight')
  return gw


def deconv3d_fwd(x, w):
  """ConvTranspose3d k3 s2 p1 op1: x (B,Cin,D,H,W), w (Cin,Cout,3,3,3) -> (B,Cout,2D,2H,2W) (mode_disparity.py:23, 25)."""
  require_gpu(x, w)
  x, w = x.contiguous(), w.contiguous()
  require_f32c(x, w)
  B, Cin, D, H, W = x.shape
  Cout = w.shape[1]
  if w.shape[0] != Cin or tuple(w.shape[2:]) != (3, 3, 3):
    raise RuntimeError('deconv3d: weight %s does not match input channels %d / kernel 3' % (tuple(w.shape), Cin))
  y = torch.empty((B, Cout, 2 * D, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
  flops = 2 * x.numel() * Cout * 27
  with torch.cuda.device_of(x), profiling.region('deconv3d_fwd', 4 * (x.numel() + y.numel() + w.numel()), flops, x.device):
    wp = _wpack3d(Cin, Cout, x.device)
    if CONV_ARITH == 'bf16x6' and lib().mode_deconv3d_split_supported(Cin, Cout) == 1 and _deconv_split_fits(D * H * W, Cout):
      check(lib().mode_deconv3d_fwd_split(ptr(x), ptr(w), ptr(y), ptr(wp), B, Cin, D, H, W, Cout, stream_of(x)), 'mode_deconv3d_fwd_split')
    else:
      check(lib().mode_deconv3d_fwd(ptr(x), ptr(w), ptr(y), ptr(wp), B, Cin, D, H, W, Cout, stream_of(x)), 'mode_deconv3d_fwd')
  return y


class Conv3dFunction(torch.autograd.Function):
  """k3 p1 convolution (stride 1|2) on the HIP kernels; autograd of F.conv3d(x, w, None, stride, 1)."""

  @staticmethod
  def forward(ctx, x, w, stride, carrier=None):
    ctx.save_for_backward(x, w)
    ctx.stride = stride
    ctx.carrier = carrier  # GradCarrier of x (x has one other consumer), or None
    ctx.amax = None
    if stride == 1 and x.is_cuda and CONV3D_S1_F16 and _split3d(x.shape[1], w.shape[0], 1, False):
      ax = known_abs_max(x)  # left by the BatchNorm pass that wrote x, where there is one
      ctx.amax = (ax if ax is not None else abs_max(x.contiguous()), _weight_abs_max(w.contiguous()))  # the backward reads both tensors again
    return conv3d_fwd(x, w, stride, amax=ctx.amax)

  @staticmethod
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    gx = None
    fwd_amax = getattr(ctx, 'amax', None)
    ag = None
    if fwd_amax is not None:  # fp16 arithmetic: gy's maximum once, for both gradients
      ag = known_abs_max(gy)  # left by the BatchNorm backward that wrote gy, where that is where gy comes from
      gy = gy.contiguous()
      if ag is None:
        ag = abs_max(gy)
    if ctx.needs_input_grad[0]:
      carrier = getattr(ctx, 'carrier', None)  # (Conv3dStatsFunction shares this backward and has none)
      prev = carrier.take(ctx) if carrier is not None else None  # the other consumer's gradient, when it came first
      gx = conv3d_bwd_data(gy, w, x.shape, ctx.stride, acc=prev, amax=(ag, fwd_amax[1]) if ag is not None else None)
      if prev is None and carrier is not None and carrier.leave(gx, ctx):
        gx = None  # first of the two: the other consumer's backward returns the sum
    gw = None
    if ctx.needs_input_grad[1]:
      sink = grad_sink(w)
      gw = conv3d_bwd_weight(gy, x, ctx.stride, into=sink, amax=(ag, fwd_amax[0]) if ag is not None else None)
      if sink is not None:
        gw = None
    return gx, gw, None, None


class Deconv3dFunction(torch.autograd.Function):
  """ConvTranspose3d k3 s2 p1 op1; its input gradient is a stride-2 convolution with the same weights and its weight
  gradient the stride-2 weight-gradient kernel with the roles of input and output gradient exchanged."""

  @staticmethod
  def forward(ctx, x, w):
    ctx.save_for_backward(x, w)
    return deconv3d_fwd(x, w)

  @staticmethod
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    gx = conv3d_fwd(gy, w, 2) if ctx.needs_input_grad[0] else None  # w (Cin,Cout,27) read as (Co=Cin, Ci=Cout)
    gw = None
    if ctx.needs_input_grad[1]:
      sink = grad_sink(w)
      gw = conv3d_bwd_weight(x, gy, 2, into=sink)  # -> (Cin, Cout, 3,3,3)
      if sink is not None:
        gw = None
    return gx, gw


def conv3d(x, w, stride=1, carrier=None):
  if carrier is not None:
    carrier.arm(x.requires_grad and torch.is_grad_enabled())
  return Conv3dFunction.apply(x, w, stride, carrier)


# BatchNorm statistics in the epilogue of the stride-1 split 3-D convolution (training) instead of a statistics pass over its output.
# Built and measured in round 4 (same box, graph-replayed step): the ten BatchNorm layers behind these convolutions get 0.41 ms
# cheaper, the convolution launches 0.36 ms dearer (0.78 -> 0.82 ms at 32 -> 32 / 48 x 256 x 128: the tile epilogue of a persistent,
# one-workgroup-per-CU kernel is not overlapped with anything, so whatever is added to it is paid in full): 73.03-73.41 against
# 73.17-73.39 ms per step.  Off by default -- the two-kernel path has one code path fewer; `bench.py --fused-bn-stats` measures it.
CONV3D_BN_STATS = False


def conv3d_stats_supported(x, w, bn):
  """Training-mode convbn_3d whose stride-1 convolution runs on the split kernel: the BatchNorm batch statistics can be taken in the
  convolution's epilogue (mode_conv3d_fwd_split_stats) instead of by a pass over its output."""
  return (CONV3D_BN_STATS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[0] > 0 and bn.training and
          bn.momentum is not None and _split3d(x.shape[1], w.shape[0], 1, False) and x.shape[0] * w.shape[0] < 65536)


class Conv3dStatsFunction(torch.autograd.Function):
  """Conv3dFunction (stride 1) that also fills `ws` -- the workspace of the BatchNorm that follows -- with the batch statistics of its
  output (partial sums per workgroup + pivots).  `ws` is a plain buffer, not differentiable."""

  @staticmethod
  def forward(ctx, x, w, ws):
    require_gpu(x, w, ws)
    x, w = x.contiguous(), w.contiguous()
    require_f32c(x, w, ws)
    B, Ci, D, H, W = x.shape
    Co = w.shape[0]
    if tuple(w.shape[1:]) != (Ci, 3, 3, 3):
      raise RuntimeError('conv3d: weight %s does not match input channels %d / kernel 3' % (tuple(w.shape), Ci))
    y = torch.empty((B, Co, D, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device_of(x), profiling.region(_tag3('conv3d_fwd', Ci, Co, 1, D, H, W), 4 * (x.numel() + y.numel() + w.numel()),
                                                   2 * y.numel() * Ci * 27, x.device):
      wp = _wpack3d(Ci, Co, x.device)
      check(lib().mode_conv3d_fwd_split_stats(ptr(x), ptr(w), ptr(y), ptr(wp), ptr(ws), B, Ci, D, H, W, Co, stream_of(x)),
            'mode_conv3d_fwd_split_stats')
    ctx.save_for_backward(x, w)
    ctx.stride = 1
    return y

  @staticmethod
  def backward(ctx, gy):
    return Conv3dFunction.backward(ctx, gy)[:3]  # (x, w, ws)


def conv3d_bn_train(x, conv_weight, bn, add=None, relu=False):
  """relu?(batch_norm_train(conv3d(x, w, stride 1, padding 1)) [+ add]) with the statistics pass folded into the convolution."""
  ws = _bn_ws(conv_weight.shape[0], x.device)
  update = bn.training and bn.track_running_stats
  y = Conv3dStatsFunction.apply(x, conv_weight, ws)
  nbt = bn.num_batches_tracked if update else None
  if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64):
    nbt.add_(1)
    nbt = None
  return BnActFunction.apply(y, add, bn.weight, bn.bias, bn.running_mean if update else None, bn.running_var if update else None, bn.momentum,
                             bn.eps, relu, nbt, 1, ws)


def deconv3d(x, w):
  return Deconv3dFunction.apply(x, w)


# ------------------------------------------------------------------------------------ fused soft-argmin head
def _tag_head(name, D4, H, W):
  """Profiling label with the shape, e.g. head_bwd[48x1024x512] (nodes along the disparity axis x output height x output width)."""
  return '%s[%dx%dx%d]' % (name, D4, H, W) if profiling.ENABLED else name


def head_fwd(logits, size, with_confidence=False):
  """logits (B,1,D4,H4,W4) -> pred (B,1,H,W) [, conf (B,1,H,W)] for size = (D,H,W).
  Replaces mode_disparity.py:131-152 (+ :157-183 for the confidence map)."""
  require_gpu(logits)
  logits = logits.contiguous()
  require_f32c(logits)
  B, one, D4, H4, W4 = logits.shape
  if one != 1:
    raise ValueError('head: expected (B,1,D/4,H/4,W/4) logits, got %s' % (tuple(logits.shape),))
  D, H, W = size
  pred = torch.empty((B, 1, H, W), dtype=logits.dtype, device=logits.device)
  conf = torch.empty_like(pred) if with_confidence else None
  nbytes = 4 * (logits.numel() + pred.numel() * (2 if with_confidence else 1))
  with torch.cuda.device_of(logits), profiling.region(_tag_head('head_fwd', D4, H, W), nbytes, 0, logits.device):
    check(lib().mode_head_fwd(ptr(logits), ptr(pred), ptr(conf) if conf is not None else None, B, D4, H4, W4, D, H, W,
                              stream_of(logits)), 'mode_head_fwd')
  return (pred, conf) if with_confidence else pred


def head_bwd(logits, gpred, size):
  require_gpu(logits, gpred)
  logits, gpred = logits.contiguous(), gpred.contiguous()
  require_f32c(logits, gpred)
  B, _, D4, H4, W4 = logits.shape
  D, H, W = size
  gl = torch.empty_like(logits)
  nbytes = 4 * (2 * logits.numel() + gpred.numel())
  with torch.cuda.device_of(logits), profiling.region(_tag_head('head_bwd', D4, H, W), nbytes, 0, logits.device):
    n = lib().mode_head_bwd_workspace_bytes(B, D4, H, W)
    ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=logits.device)
    check(lib().mode_head_bwd(ptr(logits), ptr(gpred), ptr(gl), ptr(ws), B, D4, H4, W4, D, H, W, stream_of(logits)),
          'mode_head_bwd')
  return gl


class HeadFunction(torch.autograd.Function):

  @staticmethod
  def forward(ctx, logits, size):
    ctx.save_for_backward(logits)
    ctx.size = tuple(size)
    return head_fwd(logits, size)

  @staticmethod
  def backward(ctx, gpred):
    logits, = ctx.saved_tensors
    return head_bwd(logits, gpred, ctx.size), None


def head(logits, size):
  return HeadFunction.apply(logits, size)


# ------------------------------------------------------------------------------------ the three heads + the training loss, fused
# train_disparity.py:151-158: loss = 0.5 sl1(pred1[mask], gt[mask]) + 0.7 sl1(pred2[mask], ...) + sl1(pred3[mask], ...), smooth-L1 with
# mean reduction over the valid pixels.  As torch ops on the three (B, 1, H, W) maps that is ~25 small elementwise launches forward
# and as many backward; here the loss is two tiny launches (mode_smooth_l1_masked) and its gradient is formed inside the head's
# backward kernel from the forward's own prediction (mode_head_bwd_loss): nothing elementwise between the heads and the optimizer.
def head_loss_supported(logits, size):
  B, one, D4, H4, W4 = logits.shape
  D, H, W = size
  return (logits.is_cuda and logits.dtype == torch.float32 and one == 1 and
          lib().mode_head_loss_supported(B, D4, H4, W4, D, H, W) == 1)


class HeadLossFunction(torch.autograd.Function):
  """(loss, pred1, pred2, pred3) = f(cost1, cost2, cost3, gt, scale): pred_i = head(cost_i), loss = scale * sum_i w_i * sum over the
  pixels with a ground truth (gt == gt) of smooth_l1(pred_i - gt).  `scale` is a device scalar (1 / number of valid pixels over all
  ranks: data_parallel.global_valid_count).  The predictions are returned for inspection and carry no gradient."""

  @staticmethod
  def forward(ctx, c1, c2, c3, gt, scale, weights, size):
    costs = [c.contiguous() for c in (c1, c2, c3)]
    gt = gt.contiguous()
    require_gpu(*costs, gt, scale)
    require_f32c(*costs, gt)
    scale = scale.reshape(1).to(torch.float32)
    preds = [head_fwd(c, size) for c in costs]
    if tuple(gt.shape) != tuple(preds[0].shape):
      raise RuntimeError('head_loss: ground truth %s does not match the predictions %s' % (tuple(gt.shape), tuple(preds[0].shape)))
    n = gt.numel()
    loss = torch.empty(1, dtype=torch.float32, device=gt.device)
    with torch.cuda.device_of(gt), profiling.region('smooth_l1_masked', 4 * 4 * n, 0, gt.device):
      ws = torch.empty(max(lib().mode_smooth_l1_workspace_bytes(n) // 4, 1), dtype=torch.float32, device=gt.device)
      check(lib().mode_smooth_l1_masked(ptr(preds[0]), ptr(preds[1]), ptr(preds[2]), ptr(gt), float(weights[0]), float(weights[1]),
                                        float(weights[2]), ptr(scale), ptr(loss), ptr(ws), n, stream_of(gt)), 'mode_smooth_l1_masked')
    ctx.save_for_backward(*costs, *preds, gt, scale)
    ctx.weights, ctx.size = tuple(float(w) for w in weights), tuple(size)
    ctx.mark_non_differentiable(*preds)
    return (loss.reshape(()),) + tuple(preds)

  @staticmethod
  def backward(ctx, gloss, *unused):
    c1, c2, c3, p1, p2, p3, gt, scale = ctx.saved_tensors
    sg = (scale * gloss.reshape(1)).contiguous()  # upstream gradient of the loss folded into the scale (one 1-element launch)
    D, H, W = ctx.size
    grads = []
    for c, p, wt, need in zip((c1, c2, c3), (p1, p2, p3), ctx.weights, ctx.needs_input_grad[:3]):
      if not need:
        grads.append(None)
        continue
      B, _, D4, H4, W4 = c.shape
      gl = torch.empty_like(c)
      with torch.cuda.device_of(c), profiling.region(_tag_head('head_bwd', D4, H, W), 4 * (2 * c.numel() + 2 * p.numel()), 0, c.device):
        ws = torch.empty(max(lib().mode_head_bwd_workspace_bytes(B, D4, H, W) // 4, 1), dtype=torch.float32, device=c.device)
        check(lib().mode_head_bwd_loss(ptr(c), ptr(p), ptr(gt), wt, ptr(sg), ptr(gl), ptr(ws), B, D4, H4, W4, D, H, W, stream_of(c)),
              'mode_head_bwd_loss')
      grads.append(gl)
    return grads[0], grads[1], grads[2], None, None, None, None


def head_loss(costs, size, gt, scale, weights=(0.5, 0.7, 1.0)):
  """costs = (cost1, cost2, cost3) (B, 1, D/4, H/4, W/4) each -> (loss, (pred1, pred2, pred3)); head_loss_supported(costs[0], size)."""
  out = HeadLossFunction.apply(costs[0], costs[1], costs[2], gt, scale, tuple(weights), tuple(size))
  return out[0], tuple(out[1:])


# ------------------------------------------------------------------------------------ BatchNorm (+ add) (+ ReLU)
def _tag_bn(name, y):
  """Profiling label with the tensor shape (like the convolution labels), e.g. bn_train_fwd[2x32 48x256x128]."""
  return '%s[%dx%d %s]' % (name, y.shape[0], y.shape[1], 'x'.join(str(int(v)) for v in y.shape[2:])) if profiling.ENABLED else name


def _bn_ws(C, device):
  return torch.empty(lib().mode_bn_workspace_bytes(C) // 4, dtype=torch.float32, device=device)


def _written_by_kernel(*tensors):
  """The training BatchNorm kernels update running_mean / running_var / num_batches_tracked through raw pointers; torch must see those
  writes like any in-place op: the version counters move (host-side only, no launch), so that everything keyed on them -- autograd's
  saved-tensor checks, the packed-weight cache of the eval forward (_eval_wpack) -- notices."""
  live = [t for t in tensors if t is not None]
  if live:
    torch.autograd.graph.increment_version(live)
    log = getattr(_bn_tls, 'written_log', None)
    if log is not None:  # a capture is recording: the REPLAYS write these tensors too, through the same raw pointers (GraphedStep.replay)
      log.extend(live)


def written_log_begin():
  """Start recording (on this thread) the tensors the library's kernels write through raw pointers -- BatchNorm running statistics and
  batch counters; graph_step.GraphedStep calls this around a capture: every replay of the graph writes them again, with no Python in
  between to move their version counters, so replay() moves them (the packed-weight cache of the eval forward and
  GraphedStep.stale() of a captured inference otherwise miss a BatchNorm recalibration run as replayed training steps)."""
  _bn_tls.written_log = []


def written_log_end():
  log, _bn_tls.written_log = getattr(_bn_tls, 'written_log', None) or [], None
  seen, out = set(), []
  for t in log:
    if id(t) not in seen:
      seen.add(id(t))
      out.append(t)
  return out


def _bcs(t):
  B, C = t.shape[:2]
  S = t.numel() // max(B * C, 1)
  return B, C, S


def bn_supported(y):
  B, C, S = _bcs(y)
  return y.is_cuda and y.dtype == torch.float32 and B * C < 65536 and B > 0  # (any S: rows that are not multiples of 16 bytes take the kernels' scalar path)


_bn_tls = threading.local()  # out_amax: the device scalar the last BnActFunction.forward of this thread asked its kernel to fill


def known_abs_max(t):
  """The largest-magnitude scalar the producer of t left next to it (BatchNorm's apply pass, bn_act), if t has not been written since."""
  tag = getattr(t, '_mode_amax', None)
  return tag[0] if tag is not None and tag[1] == t._version and tag[2] == t.data_ptr() else None


class BnActFunction(torch.autograd.Function):
  """out = relu?(batch_norm_train(y) [+ add]); running statistics updated in place (torch semantics)."""

  @staticmethod
  def forward(ctx, y, add, gamma, beta, running_mean, running_var, momentum, eps, relu, num_batches_tracked=None, groups=1, prestats_ws=None,
              add_carrier=None):
    ctx.add_carrier = add_carrier  # GradCarrier of `add` (the skip tensor has one other consumer), or None
    require_gpu(y, add, gamma, beta)
    y = y.contiguous()
    add = add.contiguous() if add is not None else None
    require_f32c(y, gamma, beta)
    B, C, S = _bcs(y)
    out = torch.empty_like(y)
    if B % groups:
      raise RuntimeError('BatchNorm statistics groups: batch %d is not divisible by %d' % (B, groups))
    mean = torch.empty(groups * C, dtype=torch.float32, device=y.device)
    invstd = torch.empty_like(mean)
    # ReLU without a residual: the backward rebuilds the mask from y and these coefficients instead of reading `out`
    from_y = bool(relu) and add is None
    coef = torch.empty((2, groups * C), dtype=torch.float32, device=y.device) if from_y else None
    nbytes = 4 * y.numel() * (3 + (1 if add is not None else 0))
    with torch.cuda.device_of(y), profiling.region(_tag_bn('bn_train_fwd', y), nbytes, 0, y.device):
      common = (ptr(y), ptr(add) if add is not None else None, ptr(gamma), ptr(beta), ptr(running_mean) if running_mean is not None else None,
                ptr(running_var) if running_var is not None else None, ptr(num_batches_tracked) if num_batches_tracked is not None else None,
                float(momentum), float(eps), int(relu), ptr(out), ptr(mean), ptr(invstd), ptr(coef[0]) if from_y else None,
                ptr(coef[1]) if from_y else None)
      _bn_tls.out_amax = out_amax = None
      if CONV_ARITH == 'bf16x6' and ((CONV3D_S1_F16 and y.dim() == 5) or (SPHERE_FWD_F16 and y.dim() == 4 and C % 16 == 0)):
        # the 3-D stack's activations feed stride-1 convolutions on the fp16 arithmetic, the extractor's the spherical layers: their
        # maximum comes out of this pass (the `_amax` entries; bn_act tags `out` with the buffer)
        _bn_tls.out_amax = out_amax = torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=y.device)
      if prestats_ws is not None:  # the producing convolution left the statistics in the workspace (conv3d_bn_train): no statistics pass
        if groups != 1:
          raise RuntimeError('BatchNorm with precomputed statistics takes one statistics group')
        check(lib().mode_bn_train_fwd_prestats_amax(*common, ptr(prestats_ws), lib().mode_conv3d_fwd_split_stats_partials(), B, C, S,
                                                    ptr(out_amax) if out_amax is not None else None, stream_of(y)),
              'mode_bn_train_fwd_prestats_amax')
      else:
        ws = _bn_ws(C * groups, y.device)
        check(lib().mode_bn_train_fwd_amax(*common, ptr(ws), B, C, S, groups, ptr(out_amax) if out_amax is not None else None, stream_of(y)),
              'mode_bn_train_fwd_amax')
    _written_by_kernel(running_mean, running_var, num_batches_tracked)
    ctx.save_for_backward(y, out if (relu and not from_y) else None, gamma, beta, mean, invstd, coef)
    ctx.relu, ctx.has_add, ctx.groups = bool(relu), add is not None, groups
    return out

  @staticmethod
  def backward(ctx, gout):
    y, out, gamma, beta, mean, invstd, coef = ctx.saved_tensors
    gout = gout.contiguous()
    B, C, S = _bcs(y)
    gy = torch.empty_like(y)
    need_gadd = ctx.has_add and ctx.relu and ctx.needs_input_grad[1]
    gadd = torch.empty_like(y) if need_gadd else None
    sink_g, sink_b = grad_sink(gamma), grad_sink(beta)
    fused = sink_g is not None and sink_b is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
    ggamma = sink_g if fused else torch.empty_like(gamma)
    gbeta = sink_b if fused else torch.empty_like(gamma)
    nbytes = 4 * y.numel() * (2 * (2 + (1 if out is not None else 0)) + 1 + (1 if need_gadd else 0))
    gy_amax = None
    if CONV_ARITH == 'bf16x6' and ((CONV3D_S1_F16 and y.dim() == 5) or (SPHERE_BWD_F16 and y.dim() == 4 and C % 16 == 0)):
      gy_amax = torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=y.device)  # the convolution in front reads gy in both of its gradients
    with torch.cuda.device_of(y), profiling.region(_tag_bn('bn_train_bwd', y), nbytes, 0, y.device):
      ws = _bn_ws(C * ctx.groups, y.device)
      check(lib().mode_bn_train_bwd_amax(ptr(gout), ptr(y), ptr(out) if out is not None else None, ptr(gamma), ptr(mean), ptr(invstd),
                                         ptr(coef[0]) if coef is not None else None, ptr(coef[1]) if coef is not None else None, int(ctx.relu),
                                         ptr(gy), ptr(gadd) if gadd is not None else None, ptr(ggamma), ptr(gbeta),
                                         int(fused), ptr(ws), B, C, S, ctx.groups, ptr(gy_amax) if gy_amax is not None else None,
                                         stream_of(y)), 'mode_bn_train_bwd_amax')
    if gy_amax is not None:
      gy._mode_amax = (gy_amax, gy._version, gy.data_ptr())  # (survives the engine's hand-over when the tensor's Python object does;
    if fused:                                               #  Conv3dFunction.backward computes the maximum itself otherwise)
      ggamma = gbeta = None
    if ctx.has_add and not ctx.relu:
      gadd = gout  # the add passes the gradient through unchanged
    if gadd is not None and ctx.add_carrier is not None:
      prev = ctx.add_carrier.take(ctx)
      if prev is not None:
        gadd = gadd + prev  # (the other consumer came first: this backward cannot add inside its kernel)
      elif ctx.add_carrier.leave(gadd, ctx):
        gadd = None  # the skip's other consumer adds it inside its input-gradient kernel
    return gy, gadd, ggamma, gbeta, None, None, None, None, None, None, None, None, None


def bn_eval(y, add, gamma, beta, running_mean, running_var, eps, relu):
  require_gpu(y, add, gamma, beta, running_mean, running_var)
  y = y.contiguous()
  add = add.contiguous() if add is not None else None
  require_f32c(y, gamma, beta, running_mean, running_var)
  B, C, S = _bcs(y)
  out = torch.empty_like(y)
  nbytes = 4 * y.numel() * (2 + (1 if add is not None else 0))
  with torch.cuda.device_of(y), profiling.region('bn_eval_fwd', nbytes, 0, y.device):
    check(lib().mode_bn_eval_fwd(ptr(y), ptr(add) if add is not None else None, ptr(gamma), ptr(beta), ptr(running_mean),
                                 ptr(running_var), float(eps), int(relu), ptr(out), B, C, S, stream_of(y)), 'mode_bn_eval_fwd')
  return out


def bn_act(bn, y, add=None, relu=False, groups=1, add_carrier=None):
  """groups > 1: batch statistics per group of B / groups consecutive samples, as `groups` consecutive calls would take them.
  nn.BatchNorm2d/3d `bn` applied to y, then the optional residual add and ReLU, in one fused pass (two in training).
  Same state handling as nn.BatchNorm: train mode uses batch statistics and updates running_mean / running_var /
  num_batches_tracked; eval mode uses the running statistics."""
  use_batch_stats = bn.training or bn.running_mean is None
  if use_batch_stats:
    momentum = bn.momentum
    update = bn.training and bn.track_running_stats
    nbt = bn.num_batches_tracked if update else None
    if groups > 1 and momentum is None:
      raise NotImplementedError('grouped BatchNorm statistics need a fixed momentum')
    if nbt is not None and (momentum is None or not (nbt.is_cuda and nbt.dtype == torch.int64)):
      nbt.add_(groups)  # cumulative moving average needs the count on the host; otherwise the kernel counts the batches
      if momentum is None:
        momentum = 1.0 / float(nbt)
      nbt = None
    if add_carrier is not None:
      add_carrier.arm(add is not None and add.requires_grad and torch.is_grad_enabled())
    out = BnActFunction.apply(y, add, bn.weight, bn.bias, bn.running_mean if update else None, bn.running_var if update else None,
                              momentum if momentum is not None else 0.0, bn.eps, relu, nbt, groups, None, add_carrier)
    am = getattr(_bn_tls, 'out_amax', None)
    if am is not None:
      _bn_tls.out_amax = None
      out._mode_amax = (am, out._version, out.data_ptr())  # (a later in-place write to `out` invalidates it: known_abs_max checks)
    return out
  if torch.is_grad_enabled() and (y.requires_grad or bn.weight.requires_grad):
    # eval-mode BN inside a graph that needs gradients: rare (the reference never does it); vendor ops keep autograd correct
    out = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    out = out + add if add is not None else out
    return torch.relu(out) if relu else out
  return bn_eval(y, add, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)


# ------------------------------------------------------------------------------------ classifier head in training, fused
# classifN = Sequential(convbn_3d(32, 32), ReLU, Conv3d(32, 1)) (mode_disparity.py:76-80): from the first convolution's output on --
# BatchNorm (batch statistics) + ReLU + the 32 -> 1 convolution + the residual add of mode_disparity.py:128-129 -- as ONE operator whose
# activated intermediate never reaches HBM (csrc/classif_head.hip: 5 passes over the 403 MB tensor per head and step instead of 13).
CLASSIF_FUSED = True  # bench.py --no-fused-classif measures the composition of separate operators


def classif_fused_supported(y, bn, conv):
  """y = output of the first convolution; bn = its BatchNorm3d (training mode); conv = the single-channel Conv3d behind the ReLU."""
  return (CLASSIF_FUSED and y.is_cuda and y.dtype == torch.float32 and y.dim() == 5 and y.shape[0] > 0 and y.shape[1] <= 32 and
          bn.training and bn.momentum is not None and bn.weight is not None and bn.bias is not None and
          type(conv) is torch.nn.Conv3d and conv.out_channels == 1 and conv.in_channels == y.shape[1] and conv.bias is None and
          tuple(conv.kernel_size) == (3, 3, 3) and tuple(conv.stride) == (1, 1, 1) and tuple(conv.padding) == (1, 1, 1) and
          tuple(conv.dilation) == (1, 1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros' and
          y.shape[1] * y.shape[2] * y.shape[3] * y.shape[4] < 2**30)


class ClassifHeadFunction(torch.autograd.Function):
  """cost = conv3d(relu(batch_norm_train(y)), w) [+ add]; running statistics updated in place (torch semantics)."""

  @staticmethod
  def forward(ctx, y, w, gamma, beta, add, running_mean, running_var, momentum, eps, num_batches_tracked):
    require_gpu(y, w, gamma, beta, add)
    y, w = y.contiguous(), w.contiguous()
    add = add.contiguous() if add is not None else None
    require_f32c(y, w, gamma, beta)
    B, C, D, H, W = y.shape
    cost = torch.empty((B, 1, D, H, W), dtype=y.dtype, device=y.device)
    if add is not None:
      require_f32c(add)
      if tuple(add.shape) != tuple(cost.shape):
        raise RuntimeError('classifier head: residual %s does not match the output %s' % (tuple(add.shape), tuple(cost.shape)))
    saved = torch.empty((4, C), dtype=torch.float32, device=y.device)  # mean, invstd, scale, shift
    with torch.cuda.device_of(y), profiling.region(_tag3('classif_fwd', C, 1, 1, D, H, W), 4 * (2 * y.numel() + cost.numel() * (2 if add is not None else 1)),
                                                   2 * cost.numel() * C * 27, y.device):
      ws = torch.empty(max(lib().mode_classif_workspace_bytes(B, C, D, H, W) // 4, 1), dtype=torch.float32, device=y.device)
      check(lib().mode_classif_train_fwd(ptr(y), ptr(gamma), ptr(beta), ptr(running_mean) if running_mean is not None else None,
                                         ptr(running_var) if running_var is not None else None,
                                         ptr(num_batches_tracked) if num_batches_tracked is not None else None, float(momentum), float(eps),
                                         ptr(w), ptr(add) if add is not None else None, ptr(cost), ptr(saved[0]), ptr(saved[1]), ptr(saved[2]),
                                         ptr(saved[3]), ptr(ws), B, C, D, H, W, stream_of(y)), 'mode_classif_train_fwd')
    _written_by_kernel(running_mean, running_var, num_batches_tracked)
    ctx.save_for_backward(y, w, gamma, beta, saved)
    ctx.has_add = add is not None
    return cost

  @staticmethod
  def backward(ctx, gcost):
    y, w, gamma, beta, saved = ctx.saved_tensors
    gcost = gcost.contiguous()
    B, C, D, H, W = y.shape
    gy = torch.empty_like(y)
    sink_w = grad_sink(w) if ctx.needs_input_grad[1] else None
    sink_g, sink_b = grad_sink(gamma), grad_sink(beta)
    fused = sink_w is not None and sink_g is not None and sink_b is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
    gw = sink_w if fused else torch.empty_like(w)
    ggamma = sink_g if fused else torch.empty_like(gamma)
    gbeta = sink_b if fused else torch.empty_like(beta)
    with torch.cuda.device_of(y), profiling.region(_tag3('classif_bwd', C, 1, 1, D, H, W), 4 * (3 * y.numel() + 4 * gcost.numel()),
                                                   2 * 3 * gcost.numel() * C * 27, y.device):
      ws = torch.empty(max(lib().mode_classif_workspace_bytes(B, C, D, H, W) // 4, 1), dtype=torch.float32, device=y.device)
      gy_amax = None
      if CONV3D_S1_F16 and CONV_ARITH == 'bf16x6':  # the 32 -> 32 convolution in front reads gy in both of its gradients (BnActFunction.backward)
        gy_amax = torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=y.device)
      check(lib().mode_classif_train_bwd_amax(ptr(gcost), ptr(y), ptr(w), ptr(gamma), ptr(beta), ptr(saved[0]), ptr(saved[1]), ptr(saved[2]),
                                              ptr(saved[3]), ptr(gy), ptr(gw), ptr(ggamma), ptr(gbeta), int(fused), ptr(ws), B, C, D, H, W,
                                              ptr(gy_amax) if gy_amax is not None else None, stream_of(y)), 'mode_classif_train_bwd_amax')
      if gy_amax is not None:
        gy._mode_amax = (gy_amax, gy._version, gy.data_ptr())
    if fused:
      gw = ggamma = gbeta = None
    return gy, gw, ggamma, gbeta, (gcost if ctx.has_add else None), None, None, None, None, None


def classif_head_train(y, bn, conv, add=None):
  """relu(bn(y)) -> conv (32 -> 1) [+ add] in training mode, fused (classif_fused_supported(y, bn, conv) must hold)."""
  update = bn.training and bn.track_running_stats
  nbt = bn.num_batches_tracked if update else None
  if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64):
    nbt.add_(1)
    nbt = None
  return ClassifHeadFunction.apply(y, conv.weight, bn.weight, bn.bias, add, bn.running_mean if update else None,
                                   bn.running_var if update else None, bn.momentum, bn.eps, nbt)


# ------------------------------------------------------------------------------------ fusion network: pooling, 2x2 transposed conv, head
class MaxPool2x2Function(torch.autograd.Function):
  """nn.MaxPool2d(2, stride=2) (mode_fusion.py:146, :161, :190) on csrc/fusion_ops.hip; the backward recomputes the picks from x."""

  @staticmethod
  def forward(ctx, x):
    require_gpu(x)
    x = x.contiguous()
    require_f32c(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
    with torch.cuda.device_of(x), profiling.region('maxpool2x2_fwd', 4 * (x.numel() + y.numel()), 0, x.device):
      check(lib().mode_maxpool2x2_fwd(ptr(x), ptr(y), B * C, H, W, stream_of(x)), 'mode_maxpool2x2_fwd')
    ctx.save_for_backward(x)
    return y

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, = ctx.saved_tensors
    gy = gy.contiguous()
    B, C, H, W = x.shape
    gx = torch.empty_like(x)
    with torch.cuda.device_of(x), profiling.region('maxpool2x2_bwd', 4 * (2 * x.numel() + gy.numel()), 0, x.device):
      check(lib().mode_maxpool2x2_bwd(ptr(x), ptr(gy), ptr(gx), B * C, H, W, stream_of(x)), 'mode_maxpool2x2_bwd')
    return gx


def maxpool2x2_supported(x, pool):
  ks = pool.kernel_size if isinstance(pool.kernel_size, tuple) else (pool.kernel_size,) * 2
  st = pool.stride if isinstance(pool.stride, tuple) else (pool.stride,) * 2
  pad = pool.padding if isinstance(pool.padding, tuple) else (pool.padding,) * 2
  dil = pool.dilation if isinstance(pool.dilation, tuple) else (pool.dilation,) * 2
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and ks == (2, 2) and st == (2, 2) and pad == (0, 0) and dil == (1, 1) and
          not pool.ceil_mode and not pool.return_indices and x.shape[2] >= 2 and x.shape[3] >= 2)


def maxpool2x2(x):
  return MaxPool2x2Function.apply(x)


def _deconv2x2_gemm(x, w):
  """y4 (B, 4 Co, H, W) = the 1x1 convolution half of ConvTranspose2d(Ci, Co, 2, 2): row 4 o + 2 i + j of the weight's own (Ci, 4 Co)
  storage -- mode_conv1x1_bwd_data reads a (rows-of-the-reduction, output) weight exactly as w lies."""
  B, Ci, H, W = x.shape
  R = w.shape[1] * 4
  y4 = torch.empty((B, R, H, W), dtype=x.dtype, device=x.device)
  with torch.cuda.device_of(x), profiling.region(_tag1('deconv2x2_gemm', Ci, R, 1, H, W), 4 * (x.numel() + y4.numel() + w.numel()),
                                                 2 * y4.numel() * Ci, x.device):
    wp = torch.empty(lib().mode_conv1x1_wpack_bytes(R, Ci) // 4, dtype=torch.float32, device=x.device)
    check(lib().mode_conv1x1_bwd_data(ptr(x), ptr(w), ptr(y4), ptr(wp), B, R, H, W, Ci, 1, stream_of(x)), 'mode_conv1x1_bwd_data')
  return y4


def _depth_to_space2(y4, scale, shift, relu):
  B, R, H, W = y4.shape
  Co = R // 4
  y = torch.empty((B, Co, 2 * H, 2 * W), dtype=y4.dtype, device=y4.device)
  with torch.cuda.device_of(y4), profiling.region('depth_to_space2', 8 * y4.numel(), 0, y4.device):
    check(lib().mode_depth_to_space2(ptr(y4), ptr(scale) if scale is not None else None, ptr(shift) if shift is not None else None, ptr(y),
                                     B, Co, H, W, int(relu), stream_of(y4)), 'mode_depth_to_space2')
  return y


def deconv2x2_supported(x, conv):
  """nn.ConvTranspose2d(Ci, Co, 2, 2) of the fusion decoder (mode_fusion.py:195, :212): kernel 2, stride 2, no padding; the 1x1 GEMM's
  weight gradient loads 16-byte row pieces (W % 4 == 0)."""
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(conv) is torch.nn.ConvTranspose2d and conv.kernel_size == (2, 2) and
          conv.stride == (2, 2) and conv.padding == (0, 0) and conv.output_padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and
          x.shape[3] % 4 == 0 and 4 * max(conv.in_channels, conv.out_channels) * x.shape[2] * x.shape[3] < 2**31)


class Deconv2x2Function(torch.autograd.Function):
  """y = conv_transpose2d(x, w (Ci, Co, 2, 2), bias, stride 2): the 1x1 GEMM on csrc/conv1x1.hip + the rearrangement of fusion_ops.hip."""

  @staticmethod
  def forward(ctx, x, w, bias):
    require_gpu(x, w)
    x, w = x.contiguous(), w.contiguous()
    require_f32c(x, w)
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None
    return _depth_to_space2(_deconv2x2_gemm(x, w), None, bias.contiguous() if bias is not None else None, False)

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gy):
    x, w = ctx.saved_tensors
    gy = gy.contiguous()
    B, Ci, H, W = x.shape
    Co = w.shape[1]
    R = 4 * Co
    g4 = torch.empty((B, R, H, W), dtype=gy.dtype, device=gy.device)
    with torch.cuda.device_of(gy), profiling.region('space_to_depth2', 8 * gy.numel(), 0, gy.device):
      check(lib().mode_space_to_depth2(ptr(gy), ptr(g4), B, Co, H, W, stream_of(gy)), 'mode_space_to_depth2')
    gx = gw = gb = None
    if ctx.needs_input_grad[0]:  # gx[b, c, q] = sum_r w[c][r] g4[b, r, q]: a 1x1 convolution R -> Ci with the weight as it lies
      gx = torch.empty_like(x)
      with torch.cuda.device_of(x), profiling.region(_tag1('deconv2x2_bwd_data', R, Ci, 1, H, W), 4 * (gx.numel() + g4.numel() + w.numel()),
                                                     2 * g4.numel() * Ci, x.device):
        wp = torch.empty(lib().mode_conv1x1_wpack_bytes(R, Ci) // 4, dtype=torch.float32, device=x.device)
        check(lib().mode_conv1x1_fwd(ptr(g4), ptr(w), ptr(gx), ptr(wp), B, R, H, W, Ci, 1, stream_of(x)), 'mode_conv1x1_fwd')
    if ctx.needs_input_grad[1]:  # gw[c][r] = sum x[b, c, q] g4[b, r, q]: the 1x1 weight gradient with "gy" := x, "x" := g4
      sink = grad_sink(w)
      gw = sink if sink is not None else torch.empty_like(w)
      with torch.cuda.device_of(x), profiling.region(_tag1('deconv2x2_bwd_weight', R, Ci, 1, H, W), 4 * (x.numel() + g4.numel() + w.numel()),
                                                     2 * g4.numel() * Ci, x.device):
        ws = torch.empty(max(lib().mode_conv1x1_bwd_weight_workspace_bytes(B, R, H, W, Ci, 1) // 4, 1), dtype=torch.float32, device=x.device)
        check(lib().mode_conv1x1_bwd_weight(ptr(x), ptr(g4), ptr(gw), ptr(ws), B, R, H, W, Ci, 1, int(sink is not None), stream_of(x)),
              'mode_conv1x1_bwd_weight')
      if sink is not None:
        gw = None
    if ctx.has_bias and ctx.needs_input_grad[2]:
      gb = gy.sum((0, 2, 3))
    return gx, gw, gb


def deconv2x2(x, conv):
  return Deconv2x2Function.apply(x, conv.weight, conv.bias)


def deconv2x2_bn_eval(x, conv, bn, relu):
  """relu?(eval-mode bn(conv_transpose2d(x))) in two launches: the 1x1 GEMM, then the rearrangement with bias and BatchNorm folded into
  its per-channel affine (C-element vectors, formed on the device)."""
  x, w = x.contiguous(), conv.weight.detach().contiguous()
  require_gpu(x, w)
  require_f32c(x, w)
  scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
  bias = conv.bias.detach() if conv.bias is not None else torch.zeros_like(bn.running_mean)
  shift = (bias - bn.running_mean) * scale + bn.bias.detach()
  return _depth_to_space2(_deconv2x2_gemm(x, w), scale.contiguous(), shift.contiguous(), relu)


def conv1x1_sigmoid_supported(x, conv):
  """nn.Conv2d(C, 1, 1, bias=True) + nn.Sigmoid, the last two layers of the fusion network (mode_fusion.py:228-229)."""
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and
          conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.out_channels == 1 and conv.in_channels <= 64 and
          (x.shape[2] * x.shape[3]) % 4 == 0)


class Conv1x1SigmoidFunction(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, w, bias):
    require_gpu(x, w)
    x, w = x.contiguous(), w.contiguous()
    require_f32c(x, w)
    B, C, H, W = x.shape
    s = torch.empty((B, 1, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device_of(x), profiling.region('conv1x1_sigmoid_fwd', 4 * (x.numel() + s.numel()), 2 * x.numel(), x.device):
      check(lib().mode_conv1x1_sigmoid_fwd(ptr(x), ptr(w), ptr(bias) if bias is not None else None, ptr(s), B, C, H * W, stream_of(x)),
            'mode_conv1x1_sigmoid_fwd')
    ctx.save_for_backward(x, w, s)
    ctx.has_bias = bias is not None
    return s

  @staticmethod
  @torch.autograd.function.once_differentiable
  def backward(ctx, gs):
    x, w, s = ctx.saved_tensors
    gs = gs.contiguous()
    B, C, H, W = x.shape
    gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
    gw = torch.empty_like(w)
    gb = torch.empty(1, dtype=x.dtype, device=x.device) if ctx.has_bias else None
    with torch.cuda.device_of(x), profiling.region('conv1x1_sigmoid_bwd', 4 * (3 * x.numel() + 2 * s.numel()), 4 * x.numel(), x.device):
      ws = torch.empty(max(lib().mode_conv1x1_sigmoid_bwd_workspace_bytes(B, H * W) // 4, 1), dtype=torch.float32, device=x.device)
      check(lib().mode_conv1x1_sigmoid_bwd(ptr(x), ptr(w), ptr(s), ptr(gs), ptr(gx) if gx is not None else None, ptr(gw),
                                           ptr(gb) if gb is not None else None, 0, ptr(ws), B, C, H * W, stream_of(x)), 'mode_conv1x1_sigmoid_bwd')
    return gx, gw if ctx.needs_input_grad[1] else None, gb if (ctx.has_bias and ctx.needs_input_grad[2]) else None


def conv1x1_sigmoid(x, conv):
  return Conv1x1SigmoidFunction.apply(x, conv.weight, conv.bias)


# ------------------------------------------------------------------------------------ eval mode: convolution + folded BatchNorm
# In inference every convbn / convbn_3d / sphereConvbn block (models/submodule.py:15-22, 61-74) is ONE launch: the BatchNorm
# scale goes into the packed weights, shift / residual add / ReLU into the store of the convolution kernel (the *_bn entry points
# of include/mode_hip.h).  No BatchNorm launch is left in an eval forward, and the un-normalised convolution result never
# reaches HBM.
def bn_foldable(bn, y_like=None):
  """Eval-mode BatchNorm with running statistics and affine parameters, fp32 on the GPU."""
  return (not bn.training and bn.running_mean is not None and bn.weight is not None and bn.bias is not None and bn.weight.is_cuda and
          bn.weight.dtype == torch.float32 and not torch.is_grad_enabled())


# ------------------------------------------------------------------------------------ packed weights of the eval forward, kept per layer
# Every forward entry repacks its weights (with the folded BatchNorm scale) into the `wpack` workspace before it runs: 113 small
# launches, 0.65 ms of the 10.3 ms eval forward at one pair (profiles/r04_eval_b1_kernel_stats.txt).  In eval mode the weights do not
# change between calls, so the *_bn_eval operators keep the workspace ON THE BatchNorm MODULE of the layer (it dies with the model:
# no stale hit through a recycled address), keyed on the entry, the arithmetic and (data_ptr, _version) of the weight and of the four
# BatchNorm tensors -- an optimizer step, load_state_dict or running-statistics update bumps a version and the next call repacks (the
# library's own training kernels write the running statistics through raw pointers and bump the counters themselves: _written_by_kernel).
# Writes through `.data` are invisible to the counters: invalidate_eval_packs(module) after those.  On a hit the entry runs under
# mode_weight_pack_reuse(1) (include/mode_hip.h) and skips its pack kernels.  A hipGraph captured from a forward that hit the cache
# replays WITHOUT pack kernels: GraphedStep records the versions the hits relied on and refuses to replay once one has moved.
EVAL_PACK_CACHE = True


class _PackReuse(object):

  def __init__(self, on):
    self.on = on

  def __enter__(self):
    if self.on:
      lib().mode_weight_pack_reuse(1)

  def __exit__(self, *exc):
    if self.on:
      lib().mode_weight_pack_reuse(0)
    return False


_pack_tls = threading.local()


def frozen_packs_begin():
  """Start recording which (tensor, version) pairs the packed-weight cache vouches for on this thread (graph_step.GraphedStep calls this
  around a capture): a captured eval forward that HITS the cache has no pack kernels in its graph -- its replays read the workspaces as
  they were at capture time, so they are only valid while those versions stand."""
  _pack_tls.log = []


def frozen_packs_end():
  log, _pack_tls.log = getattr(_pack_tls, 'log', None) or [], None
  return log


def invalidate_eval_packs(module):
  """Drop every packed-weight workspace kept under `module`.  Needed only after writes torch cannot see -- `param.data.copy_()`,
  `.data.mul_()`, raw-pointer writes from foreign code -- which do not move a version counter; optimizer steps, load_state_dict, in-place
  torch ops and this library's own training kernels are noticed without it."""
  for m in module.modules():
    m.__dict__.pop('_mode_hip_packed', None)


def _eval_wpack(bn, tag, w, nfloats, device):
  """(workspace, context manager to run the entry under): a kept workspace + pack reuse on a hit, a fresh one otherwise."""
  capturing = torch.cuda.is_current_stream_capturing()
  # replicas of nn.DataParallel share the original module's __dict__ entries (a shallow copy) but hold freshly broadcast parameters at
  # recycled addresses with version 0: no cache for them (train_disparity.py:264-265 only ever runs replicas in training anyway)
  if not EVAL_PACK_CACHE or bn is None or getattr(bn, '_is_replica', False):
    return torch.empty(nfloats, dtype=torch.float32, device=device), _PackReuse(False)
  tens = (w, bn.weight, bn.bias, bn.running_mean, bn.running_var)
  key = (tag, CONV_ARITH, nfloats, str(device), tuple(w.shape)) + tuple((t.data_ptr(), t._version) if t is not None else None for t in tens)
  cache = bn.__dict__.setdefault('_mode_hip_packed', {})
  wp = cache.get(key)
  if wp is not None:
    if capturing:  # a live graph now reads this workspace: it stays for the lifetime of the layer
      bn.__dict__.setdefault('_mode_hip_packed_pinned', []).append(wp)
      log = getattr(_pack_tls, 'log', None)
      if log is not None:
        log.extend((t, t._version) for t in tens if t is not None)
    return wp, _PackReuse(True)
  wp = torch.empty(nfloats, dtype=torch.float32, device=device)
  if not capturing:  # (memory of a capture belongs to the graph's pool: not kept)
    if len(cache) >= 4:
      cache.clear()
    cache[key] = wp
  return wp, _PackReuse(False)


def _epilogue(bn, add, relu, out):
  """(ctypes struct, tensors to keep alive): `add` must have the layout of `out`."""
  keep = [bn.weight.detach().contiguous(), bn.bias.detach().contiguous(), bn.running_mean.contiguous(), bn.running_var.contiguous()]
  if add is not None:
    add = add.contiguous()
    require_gpu(add)
    require_f32c(add)
    if tuple(add.shape) != tuple(out.shape):
      raise RuntimeError('residual input %s does not match the output %s' % (tuple(add.shape), tuple(out.shape)))
    keep.append(add)
  e = BnEpilogue(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), float(bn.eps), ptr(add) if add is not None else None, int(bool(relu)))
  return e, keep


def _eval_amax_buffer(device):
  """The buffer an eval kernel leaves its output's maximum in -- or None (the plain call) when no eval layer runs on the fp16 arithmetic."""
  return torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=device) if (CONV3D_EVAL_F16 and CONV_ARITH == 'bf16x6') else None


def _optr(t):
  return ptr(t) if t is not None else None


def _tag_amax(y, ay):
  if ay is not None:
    y._mode_amax = (ay, y._version, y.data_ptr())


def conv3d_bn_eval(x, w, bn, stride=1, add=None, relu=False):
  """relu?(bn(conv3d(x, w, stride, padding 1)) [+ add]) with bn in eval mode; Co > 1."""
  require_gpu(x, w)
  x, w = x.contiguous(), w.detach().contiguous()
  require_f32c(x, w)
  B, Ci, D, H, W = x.shape
  Co = w.shape[0]
  y = torch.empty((B, Co, _out3(D, stride), _out3(H, stride), _out3(W, stride)), dtype=x.dtype, device=x.device)
  e, keep = _epilogue(bn, add, relu, y)
  flops = 2 * y.numel() * Ci * 27
  with torch.cuda.device_of(x), profiling.region(_tag3('conv3d_bn_eval', Ci, Co, stride, D, H, W), 4 * (x.numel() + y.numel() + w.numel()),
                                                 flops, x.device):
    if stride == 1 and _split3d(Ci, Co, stride, False) and CONV3D_EVAL_F16:
      # two fp16 pieces (round 6): the input's maximum is the tag the producing eval kernel left (this one: y's below), else a pass;
      # the folded weights' maximum is taken inside the call, with the pack, and lives in the kept workspace
      wp, reuse = _eval_wpack(bn, 'conv3d_fwd_split_f16_bn', w, lib().mode_conv3d_wpack_bytes(Ci, Co) // 4, x.device)
      ax = _tagged_abs_max(x)
      ay = torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=x.device)
      with reuse:
        check(lib().mode_conv3d_fwd_split_f16_bn(ptr(x), ptr(w), ptr(ax), ctypes.byref(e), ptr(y), ptr(ay), ptr(wp), B, Ci, D, H, W, Co,
                                                 stream_of(x)), 'mode_conv3d_fwd_split_f16_bn')
      y._mode_amax = (ay, y._version, y.data_ptr())
    elif stride == 1 and _split3d(Ci, Co, stride, False):
      wp, reuse = _eval_wpack(bn, 'conv3d_fwd_split', w, lib().mode_conv3d_wpack_bytes(Ci, Co) // 4, x.device)
      with reuse:
        check(lib().mode_conv3d_fwd_split(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, D, H, W, Co, stream_of(x)),
              'mode_conv3d_fwd_split')
    elif stride == 2 and _split3d(Ci, Co, stride, False) and D * H * W < 2**26:
      wp, reuse = _eval_wpack(bn, 'conv3d_fwd_s2_split', w, lib().mode_conv3d_wpack_bytes(Ci, Co) // 4, x.device)
      ay = _eval_amax_buffer(x.device)  # (the stride-1 layer behind it runs on the fp16 arithmetic: its operand maximum out of this epilogue)
      with reuse:
        check(lib().mode_conv3d_fwd_s2_split_amax(ptr(x), ptr(w), ctypes.byref(e), ptr(y), _optr(ay), ptr(wp), B, Ci, D, H, W, Co, stream_of(x)),
              'mode_conv3d_fwd_s2_split_amax')
      _tag_amax(y, ay)
    else:
      wp, reuse = _eval_wpack(bn, 'conv3d_fwd_bn s%d' % stride, w, lib().mode_conv3d_wpack_bytes(Ci, Co) // 4, x.device)
      with reuse:
        check(lib().mode_conv3d_fwd_bn(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, D, H, W, Co, stride, stream_of(x)),
              'mode_conv3d_fwd_bn')
  return y


def deconv3d_bn_eval(x, w, bn, add=None, relu=False):
  require_gpu(x, w)
  x, w = x.contiguous(), w.detach().contiguous()
  require_f32c(x, w)
  B, Cin, D, H, W = x.shape
  Cout = w.shape[1]
  y = torch.empty((B, Cout, 2 * D, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
  e, keep = _epilogue(bn, add, relu, y)
  flops = 2 * x.numel() * Cout * 27
  with torch.cuda.device_of(x), profiling.region('deconv3d_bn_eval', 4 * (x.numel() + y.numel() + w.numel()), flops, x.device):
    if CONV_ARITH == 'bf16x6' and lib().mode_deconv3d_split_bn_supported(Cin, Cout) == 1 and _deconv_split_fits(D * H * W, Cout):
      # (round 5: the split kernel with its own epilogue instantiation -- residual values of four channels requested ahead of their
      # stores; round 3 had measured it no faster than the fp32 kernel with one load next to every store)
      wp, reuse = _eval_wpack(bn, 'deconv3d_fwd_split_bn', w, lib().mode_conv3d_wpack_bytes(Cin, Cout) // 4, x.device)
      ay = _eval_amax_buffer(x.device)
      with reuse:
        check(lib().mode_deconv3d_fwd_split_bn_amax(ptr(x), ptr(w), ctypes.byref(e), ptr(y), _optr(ay), ptr(wp), B, Cin, D, H, W, Cout,
                                                    stream_of(x)), 'mode_deconv3d_fwd_split_bn_amax')
      _tag_amax(y, ay)
    else:
      wp, reuse = _eval_wpack(bn, 'deconv3d_fwd_bn', w, lib().mode_conv3d_wpack_bytes(Cin, Cout) // 4, x.device)
      with reuse:
        check(lib().mode_deconv3d_fwd_bn(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Cin, D, H, W, Cout, stream_of(x)),
              'mode_deconv3d_fwd_bn')
  return y


def conv2d_bn_eval(x, w, bn, dilation=1, add=None, relu=False):
  """Stride-1 3x3 convolution with padding = dilation (1 | 2) + eval BatchNorm (+ add) (+ ReLU)."""
  require_gpu(x, w)
  x, w = x.contiguous(), w.detach().contiguous()
  require_f32c(x, w)
  B, Ci, H, W = x.shape
  Co = w.shape[0]
  y = torch.empty((B, Co, H, W), dtype=x.dtype, device=x.device)
  e, keep = _epilogue(bn, add, relu, y)
  flops = 2 * y.numel() * Ci * 9
  with torch.cuda.device_of(x), profiling.region('conv2d_bn_eval[%d->%d d%d %dx%d]' % (Ci, Co, dilation, H, W) if profiling.ENABLED
                                                 else 'conv2d_bn_eval', 4 * (x.numel() + y.numel() + w.numel()), flops, x.device):
    if CONV_ARITH == 'bf16x6' and CONV2D_EVAL_F16 and lib().mode_conv2d_split_supported(Ci, Co, dilation, 0) == 1:
      # two fp16 pieces (round 6): the input's maximum from its producer's tag (this kernel's own, in the residual blocks) or a pass
      wp, reuse = _eval_wpack(bn, 'conv2d_fwd_split_f16_bn', w, lib().mode_conv2d_wpack_bytes(Ci, Co) // 4, x.device)
      ax = _tagged_abs_max(x)
      ay = torch.empty(BN_ABSMAX_FLOATS, dtype=torch.float32, device=x.device)
      with reuse:
        check(lib().mode_conv2d_fwd_split_f16_bn(ptr(x), ptr(w), ptr(ax), ctypes.byref(e), ptr(y), ptr(ay), ptr(wp), B, Ci, H, W, Co, dilation,
                                                 stream_of(x)), 'mode_conv2d_fwd_split_f16_bn')
      _tag_amax(y, ay)
    elif CONV_ARITH == 'bf16x6' and lib().mode_conv2d_split_supported(Ci, Co, dilation, 0) == 1:
      wp, reuse = _eval_wpack(bn, 'conv2d_fwd_split', w, lib().mode_conv2d_wpack_bytes(Ci, Co) // 4, x.device)
      with reuse:
        check(lib().mode_conv2d_fwd_split(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(x)),
              'mode_conv2d_fwd_split')
    else:
      wp, reuse = _eval_wpack(bn, 'conv2d_fwd_bn', w, lib().mode_conv2d_wpack_bytes(Ci, Co) // 4, x.device)
      with reuse:
        check(lib().mode_conv2d_fwd_bn(ptr(x), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, H, W, Co, dilation, stream_of(x)),
              'mode_conv2d_fwd_bn')
  return y


def sphere_conv_bn_eval(x, pos, w, bn, stride, groups, add=None, relu=False, transposed=False):
  """Spherical (or integer-table) convolution + eval BatchNorm (+ add) (+ ReLU).  transposed: x, add and the result are in
  plane-transposed storage (B, C, W, H) -- only with a plannable table (sphere_t_supported)."""
  require_gpu(x, pos, w)
  x, w = x.contiguous(), w.detach().contiguous()
  require_f32c(x, pos, w)
  Co, _, Kh, Kw = w.shape
  if transposed:
    B, Ci, W, H = x.shape
    y = torch.empty((B, Co, W, H), dtype=x.dtype, device=x.device)
  else:
    B, Ci, H, W = x.shape
    Ho, Wo = (H - 1) // stride[0] + 1, (W - 1) // stride[1] + 1  # (the table carries padding / dilation: 'same' geometry)
    y = None
  plan = None
  if tuple(stride) == (1, 1) and Kh * Kw == 9:
    plan = sphere_plan(pos, Kh, Kw)
    if plan is not None and not transposed:
      n_wg = sum(plan[1]) * B * groups * (-(-(Co // groups) // 128))
      if n_wg < SPHERE_FWD_MIN_WG:
        plan = None
  if transposed and plan is None:
    raise RuntimeError('sphere_conv_bn_eval: plane-transposed storage needs a plannable sampling table')
  if not transposed and not _plan_usable(plan) and tuple(stride) == (1, 1) and Kh * Kw == 9:
    post = sphere_native_t(pos, Kh, Kw)
    if post is not None and sphere_t_supported(post, w, B, groups):  # ERP-like table: NCHW storage is its transposed storage
      return sphere_conv_bn_eval(x, post, w, bn, stride, groups, add, relu, transposed=True)
  flops = 2 * B * Co * H * W // (stride[0] * stride[1]) * w[0].numel()
  name = 'sphere_conv_bn_eval[%d->%d %dx%d]' % (Ci, Co, H, W) if profiling.ENABLED else 'sphere_conv_bn_eval'
  with torch.cuda.device_of(x), profiling.region(name, 4 * (2 * x.numel() + w.numel()), flops, x.device):
    if plan is not None:
      tiles, (n0, n1, n2) = plan[:2]
      wp, reuse = _eval_wpack(bn, 'sphere_fwd_win g%d %dx%d' % (groups, H, W), w,
                              lib().mode_sphere_conv_win_wpack_bytes(Ci, Co, Kh, Kw, groups) // 4, w.device)
      if transposed:
        e, keep = _epilogue(bn, add, relu, y)
        with reuse:
          _sphere_fwd_win(ptr(x), pos, w, e, ptr(y), wp, tiles, n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups, 1, stream_of(x))
      else:  # NCHW caller: the windowed kernel on plane-transposed copies (the residual is transposed with it)
        xt, yt = transpose_planes(x), torch.empty((B, Co, W, H), dtype=x.dtype, device=x.device)
        e, keep = _epilogue(bn, transpose_planes(add.contiguous()) if add is not None else None, relu, yt)
        with reuse:
          _sphere_fwd_win(ptr(xt), pos, w, e, ptr(yt), wp, tiles, n0, n1, n2, B, Ci, H, W, Co, Kh, Kw, groups, 1, stream_of(x))
        y = transpose_planes(yt)
    else:
      y = torch.empty((B, Co, Ho, Wo), dtype=x.dtype, device=x.device)
      e, keep = _epilogue(bn, add, relu, y)
      wp, reuse = _eval_wpack(bn, 'sphere_conv_fwd_bn g%d' % groups, w, lib().mode_sphere_conv_wpack_bytes(w.shape[1] * groups, Co, Kh, Kw, groups) // 4, w.device)
      with reuse:
        check(lib().mode_sphere_conv_fwd_bn(ptr(x), ptr(pos), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, H, W, Co, Kh, Kw, stride[0],
                                            stride[1], Ho, Wo, groups, stream_of(x)), 'mode_sphere_conv_fwd_bn')
  return y


def conv2d_tabled_bn_eval(x, conv, bn, add=None, relu=False):
  """nn.Conv2d `conv` (anything but the stride-1 3x3 layers) + eval BatchNorm (+ add) (+ ReLU) on the gather-and-MAC kernel; a
  kernel with more than MAX_TAPS taps (the 7x7 stem) runs its first row bands plainly and the epilogue on the last."""
  x = x.contiguous()
  w = conv.weight.detach().contiguous()
  B, Ci, H, W = x.shape
  Co, _, kh, kw = w.shape
  stride, pad, dil = tuple(conv.stride), tuple(conv.padding), tuple(conv.dilation)
  Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
  Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
  bands = _row_bands(kh, kw)
  if len(bands) == 1:
    pos = conv2d_table(H, W, kh, kw, stride, pad, dil, x.device)
    y = _tabled_bn(x, pos, w, bn, stride, add, relu, Ho, Wo)
    return y
  # several bands (the 7x7 stem): every band would need the scale, and only the last one the shift -- not worth a second epilogue
  # form for one 3 -> 32 layer: the bands run plainly and one fused BatchNorm pass follows (a 67 MB tensor, once per image pair)
  y = torch.empty((B, Co, Ho, Wo), dtype=x.dtype, device=x.device)
  for r0, rows in bands:
    pos = conv2d_table(H, W, rows, kw, stride, pad, dil, x.device, r0)
    part = y if r0 == 0 else torch.empty_like(y)
    sphere_conv_fwd(x, pos, w[:, :, r0:r0 + rows].contiguous(), part, stride, 1)
    if r0:
      y += part
  return bn_eval(y, add, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)


def _tabled_bn(x, pos, w, bn, stride, add, relu, Ho, Wo):
  B, Ci, H, W = x.shape
  Co, _, Kh, Kw = w.shape
  y = torch.empty((B, Co, Ho, Wo), dtype=x.dtype, device=x.device)
  e, keep = _epilogue(bn, add, relu, y)
  flops = 2 * y.numel() * w[0].numel()
  name = 'sphere_conv_bn_eval[%d->%d %dx%d]' % (Ci, Co, H, W) if profiling.ENABLED else 'sphere_conv_bn_eval'
  with torch.cuda.device_of(x), profiling.region(name, 4 * (x.numel() + y.numel() + w.numel()), flops, x.device):
    wp, reuse = _eval_wpack(bn, 'sphere_conv_fwd_bn g1', w, lib().mode_sphere_conv_wpack_bytes(w.shape[1], Co, Kh, Kw, 1) // 4, w.device)
    with reuse:
      check(lib().mode_sphere_conv_fwd_bn(ptr(x), ptr(pos), ptr(w), ctypes.byref(e), ptr(y), ptr(wp), B, Ci, H, W, Co, Kh, Kw, stride[0],
                                          stride[1], Ho, Wo, 1, stream_of(x)), 'mode_sphere_conv_fwd_bn')
  return y


def cost_conv_bn_eval(ref, tgt, weight, d4, bn, relu=True):
  """relu?(bn(conv3d(cost_volume(ref, tgt, d4), weight))) in eval mode, without the volume (cost_conv + folded BatchNorm)."""
  C = ref.shape[1]
  R, T = _tap_products(ref, weight[:, :C]), _tap_products(tgt, weight[:, C:])
  require_f32c(R, T)
  B, C9, H, W = R.shape
  Co = C9 // 9
  out = torch.empty((B, Co, d4, H, W), dtype=R.dtype, device=R.device)
  e, keep = _epilogue(bn, None, relu, out)
  with torch.cuda.device_of(R), profiling.region('cost_conv_assemble_fwd', 4 * (R.numel() + T.numel() + out.numel()), 0, R.device):
    ay = _eval_amax_buffer(R.device)
    check(lib().mode_cost_conv_assemble_fwd_bn_amax(ptr(R), ptr(T), ctypes.byref(e), ptr(out), _optr(ay), B, Co, d4, H, W, stream_of(R)),
          'mode_cost_conv_assemble_fwd_bn_amax')
    _tag_amax(out, ay)
  return out
