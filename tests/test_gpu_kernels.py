"""GPU (-m gpu): parity of the hand-written HIP kernels against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest
import torch

import plain_ops
import recipe
from oracle import mode_ref, sphere_conv_ref

import mode_hip
from mode_hip import functional as HF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  # fail loudly (never skip) if the GPU tier is run on a box without a GPU or without the built library
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()


def _rand(shape, seed, scale=1.0, integer=False):
  rs = np.random.RandomState(seed)
  a = rs.randint(-4, 5, size=shape).astype(np.float32) if integer else (rs.standard_normal(shape) * scale).astype(np.float32)
  return torch.from_numpy(a)


# ------------------------------------------------------------------ cost volume (a9): bit-exact
@pytest.mark.parametrize('B,C,D4,H,W', [(2, 4, 6, 5, 16), (1, 3, 7, 2, 5), (2, 2, 4, 3, 10), (1, 32, 48, 32, 128), (0, 4, 4, 4, 8)])
def test_cost_volume_fwd_bit_exact(B, C, D4, H, W):
  ref, tgt = _rand((B, C, H, W), 1), _rand((B, C, H, W), 2)
  got = HF.cost_volume_fwd(ref.to(DEV), tgt.to(DEV), D4).cpu()
  assert torch.equal(got, mode_ref.cost_volume(ref, tgt, D4))


def test_cost_volume_fwd_golden(golden):
  z = golden('model_tiny.npz')
  got = HF.cost_volume_fwd(torch.from_numpy(z['train/fea_left']).to(DEV), torch.from_numpy(z['train/fea_right']).to(DEV), 4)
  assert np.array_equal(got.cpu().numpy(), z['train/cost'])


@pytest.mark.parametrize('B,C,D4,H,W', [(2, 4, 6, 5, 16), (1, 3, 7, 2, 5), (2, 2, 4, 3, 10), (1, 8, 48, 16, 128)])
def test_cost_volume_bwd(B, C, D4, H, W):
  # integer-valued gradients: every partial sum is exact in fp32, so any summation order must agree bit for bit
  g = _rand((B, 2 * C, D4, H, W), 3, integer=True)
  ref = torch.zeros(B, C, H, W, requires_grad=True)
  tgt = torch.zeros(B, C, H, W, requires_grad=True)
  mode_ref.cost_volume(ref, tgt, D4).backward(g)
  g_ref, g_tgt = HF.cost_volume_bwd(g.to(DEV), C)
  assert torch.equal(g_ref.cpu(), ref.grad) and torch.equal(g_tgt.cpu(), tgt.grad)
  # random gradients: fp32 round-off only
  g = _rand((B, 2 * C, D4, H, W), 4)
  ref.grad = tgt.grad = None
  mode_ref.cost_volume(ref, tgt, D4).backward(g)
  g_ref, g_tgt = HF.cost_volume_bwd(g.to(DEV), C)
  assert torch.allclose(g_ref.cpu(), ref.grad, rtol=1e-5, atol=1e-5) and torch.allclose(g_tgt.cpu(), tgt.grad, rtol=1e-5, atol=1e-5)


def test_cost_volume_full_size_properties():
  """BASELINE configs 2-4: C=32, D4=48, 256x128 -- size-independent properties + exact comparison."""
  B, C, D4, H, W = 1, 32, 48, 256, 128
  ref, tgt = _rand((B, C, H, W), 5), _rand((B, C, H, W), 6)
  cost = HF.cost_volume(ref.to(DEV).requires_grad_(True), tgt.to(DEV), D4)
  assert cost.shape == (B, 2 * C, D4, H, W)
  # checksum of checksums: sum over the volume = sum_i sum_{w>=i} (ref + tgt shifted)
  exp = sum(float(ref[..., i:].double().sum() + tgt[..., :W - i].double().sum()) for i in range(D4))
  assert abs(float(cost.double().sum()) - exp) < 1e-6 * max(1.0, abs(exp)) + 1e-3
  assert torch.equal(cost.cpu(), mode_ref.cost_volume(ref, tgt, D4))
  # linearity: cv(a*r1 + r2, .) = a*cv(r1, .) + cv(r2, .) on the ref half (exact for a power of two)
  c2 = HF.cost_volume_fwd((2 * ref).to(DEV), tgt.to(DEV), D4)
  assert torch.equal(c2[:, :C], 2 * cost[:, :C]) and torch.equal(c2[:, C:], cost[:, C:])


# ------------------------------------------------------------------ cost volume + dres0[0][0] without the volume (a9 + a10)
@pytest.mark.parametrize('B,C,Co,D4,H,W', [(2, 4, 6, 6, 5, 16), (1, 3, 5, 7, 4, 5), (1, 8, 8, 1, 3, 9), (2, 32, 32, 12, 8, 40), (1, 2, 3, 4, 1, 130)])
def test_cost_conv_equals_conv3d_of_the_cost_volume(B, C, Co, D4, H, W):
  """HF.cost_conv (18 partial 2-D products + the assembly kernel and its adjoint) against conv3d(cost_volume(...)) in fp64 on
  the CPU oracle: output and the gradients of both feature maps and of the weight; more disparities than columns, D4 = 1,
  H = 1, widths beyond one block."""
  import torch.nn.functional as F
  ref, tgt = _rand((B, C, H, W), 31), _rand((B, C, H, W), 32)
  w = _rand((Co, 2 * C, 3, 3, 3), 33, 0.2)
  ra, ta, wa = ref.double().requires_grad_(True), tgt.double().requires_grad_(True), w.double().requires_grad_(True)
  y_ref = F.conv3d(mode_ref.cost_volume(ra, ta, D4), wa, None, 1, 1)
  gy = _rand(tuple(y_ref.shape), 34)
  y_ref.backward(gy.double())
  rd, td, wd = ref.to(DEV).requires_grad_(True), tgt.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
  y = HF.cost_conv(rd, td, wd, D4)
  assert tuple(y.shape) == tuple(y_ref.shape)
  y.backward(gy.to(DEV))
  tol = 2e-6 * (2 * C * 27)
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < tol * max(1.0, float(y_ref.abs().max()))
  for got, want in ((rd.grad, ra.grad), (td.grad, ta.grad), (wd.grad, wa.grad)):
    assert (got.cpu().double() - want).abs().max() < 1e-5 * max(1.0, float(want.abs().max()))
  # deterministic
  y2 = HF.cost_conv(rd, td, wd, D4)
  assert torch.equal(y2, y)


def test_cost_conv_assembly_is_exact_on_integers():
  """With integer-valued partial products every sum is exact in fp32: the assembly kernel and its adjoint must then agree
  bit for bit with a direct evaluation of the masked sums (masks = the volume's zero triangle and the zero padding)."""
  B, Co, D, H, W = 1, 2, 5, 2, 7
  R = _rand((B, 9 * Co, H, W), 41, integer=True)
  T = _rand((B, 9 * Co, H, W), 42, integer=True)
  Rd, Td = R.to(DEV).requires_grad_(True), T.to(DEV).requires_grad_(True)
  out = HF.CostConvAssemble.apply(Rd, Td, D)
  want = torch.zeros(B, Co, D, H, W)
  for kd in range(3):
    for kw in range(3):
      t = kd * 3 + kw
      for d in range(D):
        dp = d + kd - 1
        if not 0 <= dp < D:
          continue
        for w in range(W):
          wp = w + kw - 1
          if dp <= wp < W:
            want[:, :, d, :, w] += R[:, t * Co:(t + 1) * Co, :, wp] + T[:, t * Co:(t + 1) * Co, :, wp - dp]
  assert torch.equal(out.detach().cpu(), want)
  g = _rand(tuple(want.shape), 43, integer=True)
  gR, gT = torch.zeros_like(R), torch.zeros_like(T)  # the adjoint of the loop above, term by term
  for kd in range(3):
    for kw in range(3):
      t = kd * 3 + kw
      for d in range(D):
        dp = d + kd - 1
        if not 0 <= dp < D:
          continue
        for w in range(W):
          wp = w + kw - 1
          if dp <= wp < W:
            gR[:, t * Co:(t + 1) * Co, :, wp] += g[:, :, d, :, w]
            gT[:, t * Co:(t + 1) * Co, :, wp - dp] += g[:, :, d, :, w]
  out.backward(g.to(DEV))
  assert torch.equal(Rd.grad.cpu(), gR) and torch.equal(Td.grad.cpu(), gT)


def test_cost_conv_full_size_equals_the_two_kernel_path():
  """BASELINE configs 2-4 (C = 32, D4 = 48, 256 x 128): the folded layer against cost_volume + conv3d 64 -> 32 on the GPU (both
  hand-written paths, each pinned to the fp64 oracle at small sizes above), forward and all three gradients."""
  B, C, Co, D4, H, W = 1, 32, 32, 48, 256, 128
  ref, tgt = _rand((B, C, H, W), 35).to(DEV), _rand((B, C, H, W), 36).to(DEV)
  w = _rand((Co, 2 * C, 3, 3, 3), 37, 0.05).to(DEV)
  gy = _rand((B, Co, D4, H, W), 38).to(DEV)
  out = {}
  for name in ('folded', 'two-kernel'):
    r, t, ww = ref.clone().requires_grad_(True), tgt.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = HF.cost_conv(r, t, ww, D4) if name == 'folded' else HF.conv3d(HF.cost_volume(r, t, D4), ww, 1)
    y.backward(gy)
    out[name] = (y.detach(), r.grad, t.grad, ww.grad)
  for a, b_ in zip(out['folded'], out['two-kernel']):
    assert (a - b_).abs().max() <= 2e-5 * max(1.0, float(b_.abs().max()))
  # the zero triangle of the volume: columns w < d see no target feature, so d(out)/d(tgt) vanishes there by construction --
  # perturbing tgt at column W-1 must not change out[..., d, :, w] for w < W-1-... (locality check on one element)
  t2 = tgt.clone()
  t2[:, :, :, W - 1] += 1.0
  y2 = HF.cost_conv(ref, t2, w, D4)
  assert torch.equal(y2[..., : W - 2], out['folded'][0][..., : W - 2])  # tgt[W-1] reaches w' = W-1+d' >= W-1 only: outputs w >= W-2


# ------------------------------------------------------------------ sphere conv (a7/a8)
def _sphere_case(typ, ih, iw, B, ci, co, stride, groups, seed):
  pos = mode_ref.sphere_position(ih, iw, typ)
  H, W = pos.shape[2:]
  x = _rand((B, ci, H, W), seed)
  w = _rand((co, ci // groups, 3, 3), seed + 1, 0.2)
  Ho = sphere_conv_ref.out_size(H, 3, stride, 1, 1)
  Wo = sphere_conv_ref.out_size(W, 3, stride, 1, 1)
  gy = _rand((B, co, Ho, Wo), seed + 2)
  return pos, x, w, gy


SPHERE_CASES = [
    ('ERP', 16, 32, 2, 3, 4, 1, 1),  # odd Ci: zero-padded K, single partial M tile
    ('Cassini', 32, 16, 2, 8, 8, 1, 1),
    ('ERP', 16, 32, 2, 4, 6, 2, 1),  # stride 2: table sampled at (2h, 2w)
    ('Cassini', 32, 16, 2, 4, 4, 1, 2),  # groups
    ('ERP', 10, 20, 1, 5, 7, 1, 1),  # 200 pixels: ragged last tile
    ('Cassini', 64, 32, 2, 64, 128, 1, 1),  # layer4.0.conv1 shape at the tiny model size
    ('Cassini', 64, 32, 1, 128, 128, 1, 1),  # layer4 body shape
    ('ERP', 16, 32, 1, 40, 160, 1, 1),  # Co > 128: two M groups; Ci not a multiple of 8 or 14
]


@pytest.mark.parametrize('typ,ih,iw,B,ci,co,stride,groups', SPHERE_CASES)
def test_sphere_conv_fwd_bwd(typ, ih, iw, B, ci, co, stride, groups):
  pos, x, w, gy = _sphere_case(typ, ih, iw, B, ci, co, stride, groups, 11)
  cfg = ((stride, stride), (1, 1), (1, 1), groups)
  y_ref = sphere_conv_ref.forward(x.double(), pos, w.double(), *cfg)
  gx_ref, gw_ref = sphere_conv_ref.backward(x.double(), pos, w.double(), gy.double(), *cfg)
  xd, wd, pd, gyd = x.to(DEV), w.to(DEV), pos.to(DEV), gy.to(DEV)
  y = torch.full(tuple(y_ref.shape), float('nan'), device=DEV)
  HF.sphere_conv_fwd(xd, pd, wd, y, (stride, stride), groups)
  k = ci // groups * 9
  tol = 2e-6 * k  # fp32 accumulation over K products of O(1) magnitude
  assert (y.cpu().double() - y_ref).abs().max() < tol * max(1.0, float(y_ref.abs().max()))
  gx = torch.zeros_like(xd)
  HF.sphere_conv_bwd_data(gyd, pd, wd, gx, (stride, stride), groups)
  assert (gx.cpu().double() - gx_ref).abs().max() < 2e-6 * (co * 9) * max(1.0, float(gx_ref.abs().max()))
  gw = torch.zeros_like(wd)
  HF.sphere_conv_bwd_weight(gyd, pd, xd, gw, (stride, stride), groups)
  assert (gw.cpu().double() - gw_ref).abs().max() < 1e-5 * max(1.0, float(gw_ref.abs().max()))
  # accumulate semantics (sphere_conv.py:62-64): a second call adds on top
  HF.sphere_conv_bwd_weight(gyd, pd, xd, gw, (stride, stride), groups)
  assert (gw.cpu().double() - 2 * gw_ref).abs().max() < 2e-5 * max(1.0, float(gw_ref.abs().max()))
  # bwd-weight is deterministic (fixed-order split-K reduction)
  gw2 = torch.zeros_like(wd)
  HF.sphere_conv_bwd_weight(gyd, pd, xd, gw2, (stride, stride), groups)
  gw3 = torch.zeros_like(wd)
  HF.sphere_conv_bwd_weight(gyd, pd, xd, gw3, (stride, stride), groups)
  assert torch.equal(gw2, gw3)


@pytest.mark.parametrize('ih,iw,B,ci,co,groups', [(64, 128, 1, 16, 32, 1), (128, 256, 1, 8, 32, 1), (10, 20, 2, 5, 7, 1), (33, 66, 1, 12, 40, 2),
                                                   (16, 32, 1, 40, 160, 1)])
def test_sphere_conv_window_kernels_match_gather_kernels(ih, iw, B, ci, co, groups, monkeypatch):
  """The LDS-window forward (csrc/sphere_conv_win.hip) against the general gather forward on Cassini tables: all three
  window classes (equator, near-pole, wrap-around) appear at 128x256; ragged tiles (H % 32, W % 4 != 0), groups and
  Co > 128.  Same products, different summation order: fp32 round-off only."""
  pos = mode_ref.sphere_position(ih, iw, 'Cassini').to(DEV)
  H, W = pos.shape[2:]
  plan = HF.sphere_plan(pos, 3, 3)
  assert plan is not None, 'the gnomonic Cassini table must be plannable'
  x = _rand((B, ci, H, W), 5).to(DEV)
  w = _rand((co, ci // groups, 3, 3), 6, 0.2).to(DEV)
  out = {}
  monkeypatch.setattr(HF, 'SPHERE_FWD_MIN_WG', 0)  # small cases too
  for mode in ('window', 'gather'):
    monkeypatch.setattr(HF, 'SPHERE_FWD', mode)
    y = torch.full((B, co, H, W), float('nan'), device=DEV)
    HF.sphere_conv_fwd(x, pos, w, y, (1, 1), groups)
    out[mode] = y
  assert torch.isfinite(out['window']).all()
  assert (out['window'] - out['gather']).abs().max() < 2e-6 * (ci // groups * 9) * max(1.0, float(out['gather'].abs().max()))
  y_ref = sphere_conv_ref.forward(x.cpu().double(), pos.cpu(), w.cpu().double(), (1, 1), (1, 1), (1, 1), groups)
  assert (out['window'].cpu().double() - y_ref).abs().max() < 2e-6 * (ci // groups * 9) * max(1.0, float(y_ref.abs().max()))
  # weight gradient: windowed kernel on the compact tiles + general kernel on the listed polar pixels == general kernel on all
  gy = _rand((B, co, H, W), 9).to(DEV)
  gws = {}
  for mode in ('window', 'gather'):
    monkeypatch.setattr(HF, 'SPHERE_BWD_WEIGHT', mode)
    gw = torch.zeros_like(w)
    HF.sphere_conv_bwd_weight(gy, pos, x, gw, (1, 1), groups)
    gws[mode] = gw
  scale = max(1.0, float(gws['gather'].abs().max()))
  assert (gws['window'] - gws['gather']).abs().max() < 2e-5 * scale
  _, gw_ref = sphere_conv_ref.backward(x.cpu().double(), pos.cpu(), w.cpu().double(), gy.cpu().double(), (1, 1), (1, 1), (1, 1), groups)
  assert (gws['window'].cpu().double() - gw_ref).abs().max() < 1e-5 * max(1.0, float(gw_ref.abs().max()))
  monkeypatch.setattr(HF, 'SPHERE_BWD_WEIGHT', 'window')
  gw2 = torch.zeros_like(w)
  HF.sphere_conv_bwd_weight(gy, pos, x, gw2, (1, 1), groups)
  assert torch.equal(gw2, gws['window'])  # deterministic
  # the tall-window tiles: polar kernel (default) vs the pixel-list fallback on the general kernel
  monkeypatch.setattr(HF, 'SPHERE_POLAR', False)
  gw3 = torch.zeros_like(w)
  HF.sphere_conv_bwd_weight(gy, pos, x, gw3, (1, 1), groups)
  assert (gw3 - gws['window']).abs().max() < 2e-5 * scale
  assert (gw3.cpu().double() - gw_ref).abs().max() < 1e-5 * max(1.0, float(gw_ref.abs().max()))


@pytest.mark.parametrize('typ,ih,iw,B,ci,co,groups', [('Cassini', 64, 128, 2, 24, 40, 1), ('Cassini', 36, 72, 1, 8, 12, 2), ('ERP', 32, 64, 2, 16, 16, 1)])
def test_sphere_conv_bwd_data_on_transposed_storage(typ, ih, iw, B, ci, co, groups):
  """The adjoint gather run on the plane-transposed problem (transposed table, gy and gx) gives the input gradient of the
  NCHW run: same (pixel, tap, corner) products, only the order of the entries within a row of the adjoint table can differ."""
  pos = mode_ref.sphere_position(ih, iw, typ).to(DEV)
  H, W = pos.shape[2:]
  gy = _rand((B, co, H, W), 21).to(DEV)
  w = _rand((co, ci // groups, 3, 3), 22, 0.2).to(DEV)
  g_nchw = torch.full((B, ci, H, W), float('nan'), device=DEV)
  HF.sphere_conv_bwd_data(gy, pos, w, g_nchw, (1, 1), groups, overwrite=True)
  g_t = torch.full((B, ci, H, W), float('nan'), device=DEV)
  HF.sphere_conv_bwd_data(gy, pos, w, g_t, (1, 1), groups, overwrite=True, gy_transposed=HF.transpose_planes(gy))
  assert torch.isfinite(g_t).all()
  assert (g_t - g_nchw).abs().max() < 1e-5 * max(1.0, float(g_nchw.abs().max()))
  x = _rand((B, ci, H, W), 23)
  gx_ref, _ = sphere_conv_ref.backward(x.double(), pos.cpu(), w.cpu().double(), gy.cpu().double(), (1, 1), (1, 1), (1, 1), groups)
  assert (g_t.cpu().double() - gx_ref).abs().max() < 1e-5 * max(1.0, float(gx_ref.abs().max()))


def test_sphere_conv_unplannable_table_takes_the_gather_kernels():
  """A table without spatial structure cannot be windowed: the plan says so and the general kernels run."""
  g = torch.Generator().manual_seed(3)
  H, W = 24, 20
  pos = torch.stack([torch.rand(9, H, W, generator=g) * (H + 1) - 1, torch.rand(9, H, W, generator=g) * (W + 1) - 1], 1).reshape(1, 18, H, W)
  pd = pos.to(DEV)
  assert HF.sphere_plan(pd, 3, 3) is None
  x, w = _rand((1, 6, H, W), 7), _rand((8, 6, 3, 3), 8, 0.2)
  y = torch.empty((1, 8, H, W), device=DEV)
  HF.sphere_conv_fwd(x.to(DEV), pd, w.to(DEV), y, (1, 1), 1)
  y_ref = sphere_conv_ref.forward(x.double(), pos, w.double(), (1, 1), (1, 1), (1, 1), 1)
  assert (y.cpu().double() - y_ref).abs().max() < 1e-4


@pytest.mark.parametrize('kh,kw', [(1, 3), (5, 5), (2, 2)])
def test_sphere_conv_other_kernel_sizes(kh, kw):
  """Tap counts other than 9 take the generic (non-pipelined) kernels; even sizes use the reference's centre-less tap set."""
  pos = mode_ref.sphere_position(16, 32, 'ERP', (kh, kw))
  H, W = pos.shape[2:]
  ci, co = 6, 10
  x, w = _rand((2, ci, H, W), 81), _rand((co, ci, kh, kw), 82, 0.2)
  ph, pw = (kh - 1) // 2, (kw - 1) // 2
  Ho = sphere_conv_ref.out_size(H, kh, 1, ph, 1)
  Wo = sphere_conv_ref.out_size(W, kw, 1, pw, 1)
  gy = _rand((2, co, Ho, Wo), 83)
  cfg = ((1, 1), (ph, pw), (1, 1), 1)
  y_ref = sphere_conv_ref.forward(x.double(), pos, w.double(), *cfg)
  gx_ref, gw_ref = sphere_conv_ref.backward(x.double(), pos, w.double(), gy.double(), *cfg)
  xd, wd, pd, gyd = x.to(DEV), w.to(DEV), pos.to(DEV), gy.to(DEV)
  y = torch.empty(tuple(y_ref.shape), device=DEV)
  HF.sphere_conv_fwd(xd, pd, wd, y, (1, 1), 1)
  assert (y.cpu().double() - y_ref).abs().max() < 1e-4
  gx = torch.zeros_like(xd)
  HF.sphere_conv_bwd_data(gyd, pd, wd, gx, (1, 1), 1)
  assert (gx.cpu().double() - gx_ref).abs().max() < 1e-4
  gw = torch.zeros_like(wd)
  HF.sphere_conv_bwd_weight(gyd, pd, xd, gw, (1, 1), 1)
  assert (gw.cpu().double() - gw_ref).abs().max() < 1e-3


def test_sphere_conv_bwd_data_scatter_form_matches_gather_form():
  """The atomic scatter kernel (mode_sphere_conv_bwd_data, the reference's col2im structure) is kept next to the default
  gather form; both must give the same gradient."""
  pos, x, w, gy = _sphere_case('Cassini', 64, 32, 2, 24, 40, 1, 1, 91)
  xd, wd, pd, gyd = x.to(DEV), w.to(DEV), pos.to(DEV), gy.to(DEV)
  g_gather = torch.zeros_like(xd)
  HF.sphere_conv_bwd_data(gyd, pd, wd, g_gather, (1, 1), 1)
  g_scatter = torch.zeros_like(xd)
  wp = HF._wpack(wd, 1)
  dims = HF._sc_dims(xd.shape, wd.shape, gyd.shape[2:], (1, 1), 1)
  mode_hip.check(mode_hip.lib().mode_sphere_conv_bwd_data(mode_hip.ptr(gyd), mode_hip.ptr(pd), mode_hip.ptr(wd), mode_hip.ptr(g_scatter),
                                                           mode_hip.ptr(wp), *dims, mode_hip.stream_of(gyd)), 'mode_sphere_conv_bwd_data')
  assert (g_gather - g_scatter).abs().max() < 1e-4 * max(1.0, float(g_gather.abs().max()))


@pytest.mark.parametrize('name', ['erp_s1', 'cas_s1', 'erp_s2', 'cas_g2'])
def test_sphere_conv_golden(golden, name):
  from models.basic.spherical_conv.sphere_conv import SphereConv
  z = golden('sphere_conv.npz')
  ih, iw, ci, co, s, g = [int(v) for v in z[name + '/cfg']]
  m = SphereConv(ih, iw, str(z[name + '/type']), ci, co, 3, s, 1, 1, g, False).to(DEV)
  with torch.no_grad():
    m.weight.copy_(torch.from_numpy(z[name + '/w']))
  x = torch.from_numpy(z[name + '/x']).to(DEV).requires_grad_(True)
  y = m(x)
  y.backward(torch.from_numpy(z[name + '/gy']).to(DEV))
  assert np.abs(y.detach().cpu().numpy() - z[name + '/y']).max() < 1e-4
  assert np.abs(x.grad.cpu().numpy() - z[name + '/gx']).max() < 1e-4
  assert np.abs(m.weight.grad.cpu().numpy() - z[name + '/gw']).max() < 1e-3


def test_sphere_conv_layer4_full_size():
  """The benchmark shape: 128 -> 128 channels on the (1,18,256,128) Cassini table, fp32 oracle on the CPU."""
  pos, x, w, gy = _sphere_case('Cassini', 256, 128, 1, 128, 128, 1, 1, 21)
  w = w * 0.25
  cfg = ((1, 1), (1, 1), (1, 1), 1)
  y_ref = sphere_conv_ref.forward(x, pos, w, *cfg)
  gx_ref, gw_ref = sphere_conv_ref.backward(x, pos, w, gy, *cfg)
  xd, wd, pd, gyd = x.to(DEV), w.to(DEV), pos.to(DEV), gy.to(DEV)
  y = torch.empty_like(gyd)
  HF.sphere_conv_fwd(xd, pd, wd, y, (1, 1), 1)
  assert (y.cpu() - y_ref).abs().max() < 5e-4
  gx = torch.zeros_like(xd)
  HF.sphere_conv_bwd_data(gyd, pd, wd, gx, (1, 1), 1)
  assert (gx.cpu() - gx_ref).abs().max() < 5e-4
  gw = torch.zeros_like(wd)
  HF.sphere_conv_bwd_weight(gyd, pd, xd, gw, (1, 1), 1)
  assert (gw.cpu() - gw_ref).abs().max() < 2e-3 * max(1.0, float(gw_ref.abs().max()))
  # linearity in the input (size-independent property)
  y2 = torch.empty_like(gyd)
  HF.sphere_conv_fwd(2 * xd, pd, wd, y2, (1, 1), 1)
  assert torch.equal(y2, 2 * y)


@pytest.mark.parametrize('B,D4,H4,W4', [(2, 4, 16, 8), (1, 12, 8, 16), (2, 48, 12, 32)])
def test_head_loss_fused_equals_the_torch_composition(B, D4, H4, W4):
  """HF.head_loss (mode_smooth_l1_masked + mode_head_bwd_loss: the masked smooth-L1 of train_disparity.py:151-158 formed next to the
  heads) against the same loss written with torch ops on the three predictions of HF.head: value and the gradient of every logit."""
  import torch.nn.functional as F
  D, H, W = 4 * D4, 4 * H4, 4 * W4
  costs = [(_rand((B, 1, D4, H4, W4), 70 + i) * 2).to(DEV).requires_grad_(True) for i in range(3)]
  gt = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(5)) * (D / 2)
  gt[torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(6)) < 0.1] = float('nan')
  gt[0, 0, :2, :5] = 3 * D  # |pred - gt| > 1: the linear branch of the smooth-L1
  gt = gt.to(DEV)
  assert HF.head_loss_supported(costs[0], (D, H, W))
  count = (~torch.isnan(gt)).sum().float()
  loss, preds = HF.head_loss(costs, (D, H, W), gt, count.reciprocal())
  loss.backward()
  got = [c.grad.clone() for c in costs]
  for c in costs:
    c.grad = None
  mask = ~torch.isnan(gt)
  ref = 0
  for wgt, c, p in zip((0.5, 0.7, 1.0), costs, preds):
    o = HF.head(c, (D, H, W))
    assert torch.equal(o.detach(), p) and not p.requires_grad
    ref = ref + wgt * F.smooth_l1_loss(o[mask], gt[mask])
  ref.backward()
  assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
  for g, c in zip(got, costs):
    err = float((g - c.grad).abs().max())
    assert err <= 2e-6 * max(1e-6, float(c.grad.abs().max())) + 1e-12, err


def test_head_backward_takes_the_two_kernel_form_beyond_512_columns():
  """Rows wider than one 512-thread block (configs[4]: W = 1024) keep the per-pixel kernel + the row kernel; both forms agree."""
  B, D4, H4, W4 = 1, 4, 4, 160
  D, H, W = 16, 16, 640
  logits = (_rand((B, 1, D4, H4, W4), 80) * 2).to(DEV).requires_grad_(True)
  assert not HF.head_loss_supported(logits, (D, H, W))
  go = _rand((B, 1, H, W), 81).to(DEV)
  HF.head(logits, (D, H, W)).backward(go)
  wide = logits.grad.clone()
  # the same columns as two independent halves of 320 <= 512 pixels would need different interpolation weights: compare with autograd of the
  # torch composition instead
  import plain_ops
  l2 = logits.detach().clone().requires_grad_(True)
  plain_ops.head(l2, (D, H, W)).backward(go)
  assert float((wide - l2.grad).abs().max()) <= 2e-5 * max(1.0, float(l2.grad.abs().max()))


@pytest.mark.parametrize('W4,W,one_kernel', [(8, 64, False), (16, 128, False), (64, 512, False), (100, 512, False), (32, 64, True), (16, 64, True)])
def test_head_backward_with_an_upsampling_ratio_other_than_four(W4, W, one_kernel):
  """the one-kernel backward sums at most 12 pixels per low-resolution node along w -- enough for the model's x4 up-sampling
  (8-9), not for x8 (W = 512 over W4 = 64: ~18).  Such shapes must take the two-kernel form (mode_head_loss_supported says 0) and
  give the right gradient through the public entry; x2 (32 -> 64) stays on the one-kernel form."""
  import plain_ops
  B, D4, H4 = 1, 4, 4
  D, H = 16, 16
  logits = (_rand((B, 1, D4, H4, W4), 90) * 2).to(DEV).requires_grad_(True)
  assert HF.head_loss_supported(logits, (D, H, W)) == one_kernel
  go = _rand((B, 1, H, W), 91).to(DEV)
  HF.head(logits, (D, H, W)).backward(go)
  l2 = logits.detach().clone().requires_grad_(True)
  plain_ops.head(l2, (D, H, W)).backward(go)
  assert float((logits.grad - l2.grad).abs().max()) <= 2e-5 * max(1.0, float(l2.grad.abs().max()))


def test_native_seam_signature():
  """The reference's 17/20-argument pybind entry points (sphere_conv_cuda.cpp:339-345) work as documented."""
  from models.basic.spherical_conv import sphere_conv_cuda as ext
  pos, x, w, gy = _sphere_case('ERP', 16, 32, 2, 4, 8, 1, 1, 31)
  xd, wd, pd, gyd = x.to(DEV), w.to(DEV), pos.to(DEV), gy.to(DEV)
  bias = torch.arange(8, dtype=torch.float32, device=DEV)
  out = torch.empty(2, 8, 16, 32, device=DEV)
  ext.sphere_conv_forward_cuda(xd, wd, bias, xd.new_empty(0), pd, out, xd.new_empty(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, True)
  ref = sphere_conv_ref.forward(x, pos, w, (1, 1), (1, 1), (1, 1), 1) + bias.cpu().view(1, -1, 1, 1)
  assert (out.cpu() - ref).abs().max() < 1e-4
  gi, gw, gb = torch.zeros_like(xd), torch.zeros_like(wd), torch.zeros_like(bias)
  ext.sphere_conv_backward_cuda(xd, wd, bias, xd.new_empty(0), pd, xd.new_empty(0), gi, gw, gb, gyd, 3, 3, 1, 1, 1, 1, 1, 1, 1, True)
  gx_ref, gw_ref = sphere_conv_ref.backward(x, pos, w, gy, (1, 1), (1, 1), (1, 1), 1)
  assert (gi.cpu() - gx_ref).abs().max() < 1e-4 and (gw.cpu() - gw_ref).abs().max() < 1e-3
  assert (gb.cpu() - gy.sum((0, 2, 3))).abs().max() < 1e-3
  with pytest.raises(RuntimeError, match='invalid number of input planes'):
    ext.sphere_conv_forward_cuda(xd[:, :3].contiguous(), wd, bias, None, pd, out, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, False)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-4), (torch.float16, 2e-2)])
def test_native_seam_takes_the_references_dtypes(dtype, tol):
  """AT_DISPATCH_FLOATING_TYPES_AND_HALF (sphere_conv_cuda_kernel.cu:273, 367): double and half tensors are accepted at the native seam
  (computed in fp32, converted back into the caller's buffers) and through SphereConvFunction; other dtypes raise like AT_DISPATCH."""
  from models.basic.spherical_conv import sphere_conv_cuda as ext
  from models.basic.spherical_conv.sphere_conv import sphere_conv
  pos, x, w, gy = _sphere_case('Cassini', 32, 16, 2, 4, 8, 1, 1, 33)
  xd, wd, pd, gyd = x.to(DEV, dtype), w.to(DEV, dtype), pos.to(DEV), gy.to(DEV, dtype)
  out = torch.empty(2, 8, 32, 16, device=DEV, dtype=dtype)
  ext.sphere_conv_forward_cuda(xd, wd, None, None, pd, out, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, False)
  ref = sphere_conv_ref.forward(x, pos, w, (1, 1), (1, 1), (1, 1), 1)
  assert out.dtype == dtype and (out.cpu().float() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
  gi, gw = torch.zeros_like(xd), torch.ones_like(wd)  # (accumulated into: the ones must still be there)
  ext.sphere_conv_backward_cuda(xd, wd, None, None, pd, None, gi, gw, None, gyd, 3, 3, 1, 1, 1, 1, 1, 1, 1, False)
  gx_ref, gw_ref = sphere_conv_ref.backward(x, pos, w, gy, (1, 1), (1, 1), (1, 1), 1)
  assert (gi.cpu().float() - gx_ref).abs().max() < tol * max(1.0, float(gx_ref.abs().max()))
  assert (gw.cpu().float() - 1 - gw_ref).abs().max() < 10 * tol * max(1.0, float(gw_ref.abs().max()))
  xa, wa = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
  y = sphere_conv(xa, pd, wa, None, 1, 1, 1, 1)
  y.backward(gyd)
  assert y.dtype == dtype and xa.grad.dtype == dtype and wa.grad.dtype == dtype
  assert (xa.grad.cpu().float() - gx_ref).abs().max() < tol * max(1.0, float(gx_ref.abs().max()))
  with pytest.raises(RuntimeError, match='not implemented for'):
    ext.sphere_conv_forward_cuda(xd.to(torch.bfloat16), wd.to(torch.bfloat16), None, None, pd, out.to(torch.bfloat16), None, 3, 3, 1, 1, 1, 1, 1, 1,
                                 1, False)


# ------------------------------------------------------------------ 3x3x3 convolution (a10-a12)
CONV3D_CASES = [
    (2, 4, 4, 8, 8, 8),  # narrower than one MFMA column tile
    (1, 32, 32, 6, 10, 40),  # ragged W tile
    (1, 64, 32, 4, 8, 32),  # dres0.0 channel shape
    (1, 32, 64, 4, 8, 32),  # two output-channel tiles
    (2, 20, 40, 5, 7, 33),  # nothing divides anything
    (1, 64, 64, 12, 16, 32),  # hourglass inner shape (1/16 resolution of the tiny model family)
    (1, 32, 32, 12, 64, 64),  # enough tiles for the 2x8 tile shape
]


@pytest.mark.parametrize('B,Ci,Co,D,H,W', CONV3D_CASES)
def test_conv3d_fwd_bwd(B, Ci, Co, D, H, W, arith):
  import torch.nn.functional as F
  x = _rand((B, Ci, D, H, W), 41)
  w = _rand((Co, Ci, 3, 3, 3), 42, (2.0 / (27 * Co))**0.5)
  gy = _rand((B, Co, D, H, W), 43)
  xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
  y_ref = F.conv3d(xa, wa, None, 1, 1)
  y_ref.backward(gy.double())
  xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
  y = HF.conv3d(xd, wd)
  y.backward(gy.to(DEV))
  scale = max(1.0, float(y_ref.abs().max()))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 2e-6 * Ci * 27 * scale
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-6 * Co * 27 * max(1.0, float(xa.grad.abs().max()))
  gw_scale = max(1.0, float(wa.grad.abs().max()))
  assert (wd.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * gw_scale
  # deterministic weight gradient (fixed-order split-K)
  g2 = HF.conv3d_bwd_weight(gy.to(DEV), x.to(DEV))
  assert torch.equal(g2, wd.grad)


@pytest.mark.parametrize('B,Ci,Co,D,H,W', [(2, 8, 16, 8, 8, 8), (1, 32, 64, 4, 8, 64), (1, 64, 64, 6, 12, 40), (2, 20, 40, 4, 6, 70)])
def test_conv3d_stride2(B, Ci, Co, D, H, W, arith):
  """hourglass conv1 / conv3 (mode_disparity.py:15, 19): k3 s2 p1."""
  import torch.nn.functional as F
  x = _rand((B, Ci, D, H, W), 44)
  w = _rand((Co, Ci, 3, 3, 3), 45, (2.0 / (27 * Co))**0.5)
  xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
  y_ref = F.conv3d(xa, wa, None, 2, 1)
  gy = _rand(tuple(y_ref.shape), 46)
  y_ref.backward(gy.double())
  xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
  y = HF.conv3d(xd, wd, 2)
  assert y.shape == y_ref.shape
  y.backward(gy.to(DEV))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 2e-6 * Ci * 27 * max(1.0, float(y_ref.abs().max()))
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-6 * Co * 27 * max(1.0, float(xa.grad.abs().max()))
  assert (wd.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))


@pytest.mark.parametrize('B,Ci,Co,D,H,W', [(2, 8, 16, 4, 4, 4), (1, 64, 64, 3, 8, 32), (1, 64, 32, 6, 16, 32), (2, 24, 40, 2, 5, 35)])
def test_deconv3d(B, Ci, Co, D, H, W, arith):
  """hourglass conv5 / conv6 (mode_disparity.py:23, 25): ConvTranspose3d k3 s2 p1 op1, weight (Cin, Cout, 3,3,3)."""
  import torch.nn.functional as F
  x = _rand((B, Ci, D, H, W), 47)
  w = _rand((Ci, Co, 3, 3, 3), 48, 0.1)
  xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
  y_ref = F.conv_transpose3d(xa, wa, None, 2, 1, 1)
  gy = _rand(tuple(y_ref.shape), 49)
  y_ref.backward(gy.double())
  xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
  y = HF.deconv3d(xd, wd)
  assert y.shape == y_ref.shape == (B, Co, 2 * D, 2 * H, 2 * W)
  y.backward(gy.to(DEV))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 2e-6 * Ci * 27 * max(1.0, float(y_ref.abs().max()))
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-6 * Co * 27 * max(1.0, float(xa.grad.abs().max()))
  assert (wd.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))


@pytest.mark.parametrize('B,Ci,D,H,W', [(2, 32, 6, 12, 40), (1, 20, 3, 5, 33), (1, 40, 2, 9, 64), (2, 32, 1, 1, 7)])
def test_conv3d_single_output_channel(B, Ci, D, H, W):
  """classifN[2]: Conv3d(32 -> 1) (mode_disparity.py:76-80): stencil forward, MFMA input / weight gradients with the taps
  as a GEMM dimension; ragged tiles, channel counts off the 32-wide tile, volumes thinner than the halo."""
  import torch.nn.functional as F
  x = _rand((B, Ci, D, H, W), 54)
  w = _rand((1, Ci, 3, 3, 3), 55, 0.05)
  xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
  y_ref = F.conv3d(xa, wa, None, 1, 1)
  gy = _rand(tuple(y_ref.shape), 56)
  y_ref.backward(gy.double())
  xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
  y = HF.conv3d(xd, wd, 1)
  y.backward(gy.to(DEV))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 1e-4
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 1e-4
  assert (wd.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))


# ------------------------------------------------------------------ regular 3x3 Conv2d: weight gradient (a3)
@pytest.mark.parametrize('B,Ci,Co,H,W,dil', [(2, 32, 32, 16, 64, 1), (1, 64, 64, 9, 40, 1), (2, 20, 40, 7, 33, 1), (1, 64, 64, 12, 32, 2),
                                             (2, 3, 5, 5, 70, 2), (1, 128, 128, 8, 32, 1), (1, 8, 8, 2, 3, 2), (4, 64, 64, 64, 32, 1), (1, 96, 72, 10, 64, 2)])
def test_conv2d_3x3_kernels(B, Ci, Co, H, W, dil, monkeypatch, arith):
  """mode_conv2d_bwd_weight against torch's fp64 conv2d autograd: ragged tiles, channel counts off the 32-wide block, dilation 2,
  images smaller than the halo; accumulate semantics; the autograd Function (vendor forward / input gradient + own weight
  gradient) end to end."""
  import torch.nn.functional as F
  x, w = _rand((B, Ci, H, W), 61), _rand((Co, Ci, 3, 3), 62, 0.2)
  xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
  y_ref = F.conv2d(xa, wa, None, 1, dil, dil)
  gy = _rand(tuple(y_ref.shape), 63)
  y_ref.backward(gy.double())
  scale = max(1.0, float(wa.grad.abs().max()))
  gw = HF.conv2d_bwd_weight(gy.to(DEV), x.to(DEV), dil)
  assert (gw.cpu().double() - wa.grad).abs().max() < 2e-5 * scale
  gw2 = HF.conv2d_bwd_weight(gy.to(DEV), x.to(DEV), dil)
  assert torch.equal(gw, gw2)  # deterministic
  acc = gw.clone()
  HF.conv2d_bwd_weight(gy.to(DEV), x.to(DEV), dil, into=acc)
  assert (acc.cpu().double() - 2 * wa.grad).abs().max() < 4e-5 * scale
  # own forward / input-gradient kernels (used at quarter resolution), directly
  if Co <= 128 and Ci <= 128:
    y = HF.conv2d_fwd(x.to(DEV), w.to(DEV), dil)
    assert (y.cpu().double() - y_ref.detach()).abs().max() < 2e-6 * (Ci * 9) * max(1.0, float(y_ref.abs().max()))
    gx = HF.conv2d_bwd_data(gy.to(DEV), w.to(DEV), dil)
    assert (gx.cpu().double() - xa.grad).abs().max() < 2e-6 * (Co * 9) * max(1.0, float(xa.grad.abs().max()))
  # the autograd Function end to end, on both routes (own kernels / vendor forward and input gradient)
  for own_limit in (1 << 30, 0):
    monkeypatch.setattr(HF, 'CONV2D_OWN_MAX_PIXELS', own_limit)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = HF.conv2d_3x3(xd, wd, dil)
    y.backward(gy.to(DEV))
    assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 1e-3 * max(1.0, float(y_ref.abs().max()))  # (vendor Winograd: looser)
    assert (xd.grad.cpu().double() - xa.grad).abs().max() < 1e-3 * max(1.0, float(xa.grad.abs().max()))
    assert (wd.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * scale


def test_conv2d_3x3_full_size_against_the_vendor_library():
  """The extractor's largest regular layer (64 -> 64 at 512 x 256, 4 images) and its dilated quarter-resolution layer: own
  forward / input gradient / weight gradient against the vendor library on the GPU (fp32 both sides; the vendor's Winograd
  has the larger round-off, hence 1e-3)."""
  import torch.nn.functional as F
  for (ci, co, H, W, dil) in ((64, 64, 512, 256, 1), (64, 64, 256, 128, 2)):
    x, w = _rand((4, ci, H, W), 71).to(DEV), _rand((co, ci, 3, 3), 72, 0.05).to(DEV)
    gy = _rand((4, co, H, W), 73).to(DEV)
    y = HF.conv2d_fwd(x, w, dil)
    y_v = F.conv2d(x, w, None, 1, dil, dil)
    assert (y - y_v).abs().max() <= 1e-3 * max(1.0, float(y_v.abs().max()))
    gx_v, gw_v = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [dil, dil], [dil, dil], False, [0, 0], 1, [True, True, False])[:2]
    gx = HF.conv2d_bwd_data(gy, w, dil)
    assert (gx - gx_v).abs().max() <= 1e-3 * max(1.0, float(gx_v.abs().max()))
    gw = HF.conv2d_bwd_weight(gy, x, dil)
    assert (gw - gw_v).abs().max() <= 1e-3 * max(1.0, float(gw_v.abs().max()))
    # linearity in the input (exact for a power of two)
    assert torch.equal(HF.conv2d_fwd(2 * x, w, dil), 2 * y)


# ------------------------------------------------------------------ the other regular Conv2d layers: integer-table gather-and-MAC (a3)
@pytest.mark.parametrize('B,Ci,Co,H,W,k,s,p,d', [
    (2, 3, 32, 64, 32, 7, 2, 3, 1),      # firstconv.0 (submodule.py:155)
    (2, 64, 64, 32, 16, 3, 2, 1, 1),     # layer2.0.conv1 (:158)
    (2, 64, 64, 32, 16, 1, 2, 0, 1),     # layer2.0.downsample (:167-174)
    (2, 32, 64, 16, 24, 1, 1, 0, 1),     # layer1.0.downsample
    (1, 256, 128, 8, 16, 1, 1, 0, 1),    # lastconv.0 (:162)
    (1, 128, 32, 8, 16, 1, 1, 0, 1),     # lastconv.4
    (1, 5, 7, 9, 11, 3, 2, 1, 1),        # odd sizes: the last window hangs over the border
    (1, 4, 6, 12, 10, 3, 1, 2, 2),       # dilation 2
    (1, 6, 4, 10, 13, 5, 3, 2, 1),       # 5x5 stride 3
])
def test_conv2d_on_the_integer_table(B, Ci, Co, H, W, k, s, p, d):
  """nn.Conv2d layers other than stride-1 3x3 run on the spherical operator's kernels with an integer sampling table
  (functional.conv2d_tabled): forward, input gradient and weight gradient against torch's fp64 conv2d autograd."""
  import torch.nn as nn
  import torch.nn.functional as F
  conv = nn.Conv2d(Ci, Co, k, s, p, d, bias=False)
  x = _rand((B, Ci, H, W), 81)
  with torch.no_grad():
    conv.weight.copy_(_rand((Co, Ci, k, k), 82, (2.0 / (Ci * k * k))**0.5))
  xa, wa = x.double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
  y_ref = F.conv2d(xa, wa, None, s, p, d)
  gy = _rand(tuple(y_ref.shape), 83)
  y_ref.backward(gy.double())
  conv = conv.to(DEV)
  assert HF.conv2d_tabled_supported(x.to(DEV), conv)
  xd = x.to(DEV).requires_grad_(True)
  y = HF.conv2d_tabled(xd, conv)
  assert tuple(y.shape) == tuple(y_ref.shape)
  y.backward(gy.to(DEV))
  tol = 2e-6 * (Ci * k * k)
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < tol * max(1.0, float(y_ref.abs().max()))
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-6 * (Co * k * k) * max(1.0, float(xa.grad.abs().max()))
  assert (conv.weight.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))
  # the module route: stage3d.conv3 sends the layer here (no vendor kernel in the step)
  from models import stage3d
  with torch.no_grad():
    via_module = stage3d.conv3(conv, x.to(DEV))
  if (k, s, p) == (3, 1, d) or k in (1, 7):  # layers with kernels of their own (conv2d.hip, conv1x1.hip, conv_stem.hip): same
    assert (via_module - y.detach()).abs().max() < tol * max(1.0, float(y_ref.abs().max()))  # result up to summation order
  else:
    assert torch.equal(via_module, y.detach())


# ------------------------------------------------------------------ BatchNorm + add + ReLU (a15)
# (the last three: spatial sizes that are not multiples of 4 -- rows not 16-byte aligned: the kernels' scalar path, round 4)
@pytest.mark.parametrize('shape', [(2, 8, 4, 6, 8), (1, 32, 6, 16, 32), (2, 64, 24, 32), (3, 5, 2, 2, 4), (2, 6, 3, 5, 7), (2, 16, 9, 11),
                                   (1, 8, 5, 37, 41)])
@pytest.mark.parametrize('relu,with_add', [(False, False), (True, False), (True, True), (False, True)])
def test_bn_act_train_and_eval(shape, relu, with_add):
  import torch.nn as nn
  C = shape[1]
  BN = nn.BatchNorm3d if len(shape) == 5 else nn.BatchNorm2d
  ref_bn, dev_bn = BN(C).double(), BN(C).to(DEV)
  g = torch.Generator().manual_seed(7)
  gamma = 1 + 0.2 * torch.randn(C, generator=g)
  beta = 0.3 * torch.randn(C, generator=g)
  with torch.no_grad():
    for bn in (ref_bn, dev_bn):
      bn.weight.copy_(gamma)
      bn.bias.copy_(beta)
  y = _rand(shape, 71, 2.0) + 1.5  # non-zero mean: exercises the variance cancellation
  add = _rand(shape, 72) if with_add else None
  gout = _rand(shape, 73)
  ya = y.double().requires_grad_(True)
  aa = add.double().requires_grad_(True) if with_add else None
  o_ref = ref_bn(ya)
  if with_add:
    o_ref = o_ref + aa
  if relu:
    o_ref = torch.relu(o_ref)
  o_ref.backward(gout.double())
  yd = y.to(DEV).requires_grad_(True)
  ad = add.to(DEV).requires_grad_(True) if with_add else None
  out = HF.bn_act(dev_bn, yd, ad, relu)
  out.backward(gout.to(DEV))
  assert (out.detach().cpu().double() - o_ref.detach()).abs().max() < 2e-5
  assert (yd.grad.cpu().double() - ya.grad).abs().max() < 5e-5 * max(1.0, float(ya.grad.abs().max()))
  if with_add:
    assert (ad.grad.cpu().double() - aa.grad).abs().max() < 1e-6
  assert (dev_bn.weight.grad.cpu().double() - ref_bn.weight.grad).abs().max() < 1e-4 * max(1.0, float(ref_bn.weight.grad.abs().max()))
  assert (dev_bn.bias.grad.cpu().double() - ref_bn.bias.grad).abs().max() < 1e-4 * max(1.0, float(ref_bn.bias.grad.abs().max()))
  assert (dev_bn.running_mean.cpu().double() - ref_bn.running_mean).abs().max() < 1e-5
  assert (dev_bn.running_var.cpu().double() - ref_bn.running_var).abs().max() < 1e-4
  assert int(dev_bn.num_batches_tracked) == 1
  # eval mode uses the running statistics just updated
  ref_bn.eval()
  dev_bn.eval()
  with torch.no_grad():
    e_ref = ref_bn(y.double())
    if with_add:
      e_ref = e_ref + add.double()
    if relu:
      e_ref = torch.relu(e_ref)
    e = HF.bn_act(dev_bn, y.to(DEV), add.to(DEV) if with_add else None, relu)
  assert (e.cpu().double() - e_ref).abs().max() < 2e-5


@pytest.mark.parametrize('shape,groups', [((4, 8, 12, 16), 2), ((6, 5, 4, 6, 8), 3), ((2, 64, 32, 64), 2), ((4, 6, 7, 9), 2)])
@pytest.mark.parametrize('relu,with_add', [(True, False), (True, True), (False, False)])
def test_bn_act_grouped_statistics(shape, groups, relu, with_add):
  """groups = n: exactly what n consecutive calls of the module on the n sub-batches compute -- outputs, gradients (affine
  gradients summed over the calls), running statistics updated call after call, num_batches_tracked += n."""
  import torch.nn as nn
  C = shape[1]
  BN = nn.BatchNorm3d if len(shape) == 5 else nn.BatchNorm2d
  ref_bn, dev_bn = BN(C).double(), BN(C).to(DEV)
  g = torch.Generator().manual_seed(9)
  with torch.no_grad():
    for bn in (ref_bn, dev_bn):
      bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(1)))
      bn.bias.copy_(0.3 * torch.randn(C, generator=torch.Generator().manual_seed(2)))
  y = torch.randn(shape, generator=g) * 2 + torch.arange(shape[0]).view(-1, *([1] * (len(shape) - 1))).float()  # groups differ in mean
  add = _rand(shape, 72) if with_add else None
  gout = _rand(shape, 73)
  ya = y.double().requires_grad_(True)
  aa = add.double().requires_grad_(True) if with_add else None
  outs, pre = [], []
  for part, apart in zip(ya.chunk(groups, 0), aa.chunk(groups, 0) if with_add else [None] * groups):
    o = ref_bn(part)
    if with_add:
      o = o + apart
    pre.append(o.detach())
    outs.append(torch.relu(o) if relu else o)
  o_ref = torch.cat(outs, 0)
  o_ref.backward(gout.double())
  # elements whose pre-ReLU value is zero to fp32 round-off: an fp32 evaluation may legitimately put them on the other side of the
  # ReLU than the fp64 reference (one such element per ~10^6 with these tensors); their gradient terms are excluded / allowed for
  amb = (torch.cat(pre, 0).abs() < 1e-5) if relu else torch.zeros(shape, dtype=torch.bool)
  n_amb = int(amb.sum())
  assert n_amb <= 4
  yd = y.to(DEV).requires_grad_(True)
  ad = add.to(DEV).requires_grad_(True) if with_add else None
  out = HF.bn_act(dev_bn, yd, ad, relu, groups=groups)
  out.backward(gout.to(DEV))
  assert (out.detach().cpu().double() - o_ref.detach()).abs().max() < 2e-5
  if n_amb == 0:
    assert (yd.grad.cpu().double() - ya.grad).abs().max() < 5e-5 * max(1.0, float(ya.grad.abs().max()))
  else:  # a flipped element moves the statistics terms of its whole channel by 1 / n: compare away from it, loosely
    assert ((yd.grad.cpu().double() - ya.grad).abs() * (~amb)).max() < 1e-3 * max(1.0, float(ya.grad.abs().max()))
  if with_add:
    assert ((ad.grad.cpu().double() - aa.grad).abs() * (~amb)).max() < 1e-6
  for a, b in ((dev_bn.weight.grad, ref_bn.weight.grad), (dev_bn.bias.grad, ref_bn.bias.grad)):
    assert (a.cpu().double() - b).abs().max() < 1e-4 * max(1.0, float(b.abs().max())) + 25.0 * n_amb
  assert (dev_bn.running_mean.cpu().double() - ref_bn.running_mean).abs().max() < 1e-5
  assert (dev_bn.running_var.cpu().double() - ref_bn.running_var).abs().max() < 1e-4
  assert int(dev_bn.num_batches_tracked) == groups == int(ref_bn.num_batches_tracked)


@pytest.mark.parametrize('B,Ci,Co,D,H,W', [(2, 32, 32, 6, 20, 40), (1, 32, 64, 5, 9, 33), (2, 64, 64, 4, 16, 32)])
@pytest.mark.parametrize('relu,with_add', [(True, False), (False, True)])
def test_conv3d_with_batchnorm_statistics_in_its_epilogue(B, Ci, Co, D, H, W, relu, with_add, monkeypatch):
  """Training-mode convbn_3d with the statistics pass folded into the split convolution kernel (HF.conv3d_bn_train, round 4) against
  float64: output, input / weight / affine gradients, running statistics -- and against the two-kernel path it replaces.  The input has
  a large mean (|mean| >> std after the convolution): the shifted sums must not cancel (their pivot is the layer's own first output value,
  whatever the running statistics hold)."""
  import torch.nn as nn
  if HF.CONV_ARITH != 'bf16x6':
    pytest.skip('the statistics epilogue belongs to the split kernel')
  x = _rand((B, Ci, D, H, W), 301) + 3.0
  w = _rand((Co, Ci, 3, 3, 3), 302, (2.0 / (27 * Ci))**0.5) + 0.02
  add = _rand((B, Co, D, H, W), 303) if with_add else None
  gout = _rand((B, Co, D, H, W), 304)
  conv64 = nn.Conv3d(Ci, Co, 3, 1, 1, bias=False).double()
  bn64 = nn.BatchNorm3d(Co).double()
  with torch.no_grad():
    conv64.weight.copy_(w.double())
  xa = x.double().requires_grad_(True)
  o = bn64(conv64(xa))
  o = o + add.double() if with_add else o
  o = torch.relu(o) if relu else o
  o.backward(gout.double())
  results = []
  for fused, rm in ((True, 0.0), (True, 50.0), (False, 0.0)):
    bn = nn.BatchNorm3d(Co).to(DEV)
    with torch.no_grad():
      bn.running_mean.fill_(rm)
    wd = w.to(DEV).requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    ad = add.to(DEV) if with_add else None
    if fused:
      monkeypatch.setattr(HF, 'CONV3D_BN_STATS', True)
      assert HF.conv3d_stats_supported(xd, wd, bn)
      out = HF.conv3d_bn_train(xd, wd, bn, ad, relu)
    else:
      out = HF.bn_act(bn, HF.conv3d(xd, wd, 1), ad, relu)
    out.backward(gout.to(DEV))
    results.append(out.detach())
    scale = max(1.0, float(o.detach().abs().max()))
    assert (out.detach().cpu().double() - o.detach()).abs().max() < 2e-4 * scale, (fused, rm)
    assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-4 * max(1.0, float(xa.grad.abs().max()))
    assert (wd.grad.cpu().double() - conv64.weight.grad).abs().max() < 2e-4 * max(1.0, float(conv64.weight.grad.abs().max()))
    assert (bn.weight.grad.cpu().double() - bn64.weight.grad).abs().max() < 2e-4 * max(1.0, float(bn64.weight.grad.abs().max()))
    ref_rm = (1 - 0.1) * rm + 0.1 * conv64(x.double()).transpose(0, 1).reshape(Co, -1).mean(1).detach()
    assert (bn.running_mean.cpu().double() - ref_rm).abs().max() < 1e-4 * max(1.0, float(ref_rm.abs().max()))
    assert (bn.running_var.cpu().double() - bn64.running_var).abs().max() < 1e-3 * max(1.0, float(bn64.running_var.abs().max()))
    assert int(bn.num_batches_tracked) == 1
  assert (results[0] - results[2]).abs().max() < 5e-5 * max(1.0, float(results[2].abs().max()))  # fused against the two-kernel path


# ------------------------------------------------------------------ fused head (a13/a14)
@pytest.mark.parametrize('B,D4,H4,W4,scale', [(2, 4, 6, 8, 4), (1, 12, 5, 7, 4), (1, 3, 4, 4, 3), (2, 48, 8, 16, 4)])
def test_head_fwd_bwd_conf(B, D4, H4, W4, scale):
  lg = _rand((B, 1, D4, H4, W4), 61, 3.0)
  D, H, W = D4 * scale, H4 * scale, W4 * scale
  la = lg.double().requires_grad_(True)
  pred_ref, prob = mode_ref.disparity_head(la, D, H, W, return_prob=True)
  conf_ref = mode_ref.confidence_map(pred_ref.detach(), prob.detach())
  g = _rand((B, 1, H, W), 62)
  pred_ref.backward(g.double())
  ld = lg.to(DEV).requires_grad_(True)
  pred = HF.head(ld, (D, H, W))
  pred.backward(g.to(DEV))
  assert (pred.detach().cpu().double() - pred_ref.detach()).abs().max() < 1e-4 * D
  assert (ld.grad.cpu().double() - la.grad).abs().max() < 1e-4 * max(1.0, float(la.grad.abs().max()))
  p2, conf = HF.head_fwd(lg.to(DEV), (D, H, W), with_confidence=True)
  assert torch.equal(p2, pred.detach())
  # round() of a prediction that sits within float error of x.5 may legitimately differ: compare where it is stable
  stable = ((pred_ref.detach() - pred_ref.detach().round()).abs() - 0.5).abs() > 1e-3
  assert ((conf.cpu().double() - conf_ref).abs()[stable]).max() < 1e-4


def test_head_golden(golden):
  z = golden('model_tiny.npz')
  # eval/logits3 is the classif3 module output; the head input adds cost2, so use the oracle head on the same tensor
  lg = torch.from_numpy(z['eval/logits3'])
  ref = mode_ref.disparity_head(lg, 16, 64, 32)
  got = HF.head_fwd(lg.to(DEV), (16, 64, 32))
  assert (got.cpu() - ref).abs().max() < 1e-4


def test_head_full_size_properties():
  lg = _rand((1, 1, 48, 256, 128), 63, 4.0).to(DEV)
  pred = HF.head_fwd(lg, (192, 1024, 512))
  assert pred.shape == (1, 1, 1024, 512)
  assert float(pred.min()) >= 0.0 and float(pred.max()) <= 191.0
  # softmax is shift invariant; a constant volume gives the mean disparity index
  assert (HF.head_fwd(lg + 3.0, (192, 1024, 512)) - pred).abs().max() < 2e-3
  flat = HF.head_fwd(torch.zeros_like(lg), (192, 1024, 512))
  assert (flat - 95.5).abs().max() < 1e-3
  # Against the float64 oracle on a crop of the same volume (the whole volume needs ~1.6 GB and tens of seconds on the CPU), next to
  # the composition of separate vendor ops in fp32 -- what the reference runs (F.upsample + F.softmax + disparityregression).  On random
  # logits of this scale the softmax is multi-modal with modes ~100 px apart, so ANY fp32 evaluation is off by up to ~7e-4 px (a relative
  # error of 3e-6 in an exponential x the distance between the modes); the fused kernel must not be further from float64 than that.
  crop = lg[:, :, :, 96:128, 32:64].contiguous()
  truth = mode_ref.disparity_head(crop.cpu().double(), 192, 128, 128)
  own = (HF.head_fwd(crop, (192, 128, 128)).cpu().double() - truth).abs()
  vend = (plain_ops.head(crop, (192, 128, 128)).cpu().double() - truth).abs()
  print('head at benchmark depth, random logits x4: |hip - fp64| max %.2e mean %.2e; vendor fp32 composition: max %.2e mean %.2e' %
        (own.max(), own.mean(), vend.max(), vend.mean()))
  assert float(own.max()) <= max(1e-3, 1.5 * float(vend.max())) and float(own.mean()) <= max(5e-5, 1.5 * float(vend.mean()))
  ref = plain_ops.head(lg, (192, 1024, 512))
  d = (ref - pred).abs()
  print('whole volume, fused kernel against the vendor composition (two fp32 evaluations): max %.2e mean %.2e' % (float(d.max()), float(d.mean())))
  assert float(d.max()) < 1e-2 and float(d.mean()) < 2e-4


def test_conv3d_full_size_vs_vendor():
  """dres-type layer at the benchmark volume (32->32 @ 48x256x128): compare with the vendor library on the same GPU
  (both fp32; an fp64 CPU run of this size takes minutes) and check linearity."""
  import torch.nn.functional as F
  x = _rand((1, 32, 48, 256, 128), 51).to(DEV)
  w = _rand((32, 32, 3, 3, 3), 52, (2.0 / (27 * 32))**0.5).to(DEV)
  y = HF.conv3d_fwd(x, w)
  ref = F.conv3d(x, w, None, 1, 1)
  assert (y - ref).abs().max() < 2e-4 * max(1.0, float(ref.abs().max()))
  assert torch.equal(HF.conv3d_fwd(2 * x, w), 2 * y)
  # the corner voxel only sees the 2x2x2 in-range part of the kernel (zero padding)
  corner = torch.einsum('cdhw,ocdhw->o', x[0, :, :2, :2, :2].cpu().double(), w[:, :, 1:, 1:, 1:].cpu().double())
  assert (y[0, :, 0, 0, 0].cpu().double() - corner).abs().max() < 1e-4


# ------------------------------------------------------------------ eval mode: convolution + folded BatchNorm in one launch (a15)
def _eval_bn(C, seed, dims=3):
  import torch.nn as nn
  bn = (nn.BatchNorm3d if dims == 3 else nn.BatchNorm2d)(C).to(DEV).eval()
  with torch.no_grad():
    bn.weight.copy_(_rand((C,), seed) * 0.2 + 1.0)
    bn.bias.copy_(_rand((C,), seed + 1) * 0.3)
    bn.running_mean.copy_(_rand((C,), seed + 2) * 0.5)
    bn.running_var.copy_(_rand((C,), seed + 3).abs() + 0.3)
  return bn


def _unfused(bn, y, add, relu):
  y = torch.nn.functional.batch_norm(y.double(), bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(),
                                     False, 0.0, bn.eps)
  if add is not None:
    y = y + add.double()
  return torch.relu(y) if relu else y


FOLD_VARIANTS = [(True, False), (False, True), (True, True), (False, False)]


@pytest.mark.parametrize('relu,with_add', FOLD_VARIANTS)
def test_folded_batchnorm_conv3d(relu, with_add, arith):
  """mode_conv3d_fwd_bn (stride 1 with one and two output-channel tiles, stride 2) and mode_deconv3d_fwd_bn -- and, in bf16x6 mode, the
  split kernels with the same epilogue (mode_conv3d_fwd_split, mode_conv3d_fwd_s2_split) -- against
  convolution -> fp64 eval BatchNorm (+ add) (+ ReLU), under torch.no_grad (the fused form is inference only)."""
  import torch.nn.functional as F
  with torch.no_grad():
    for (ci, co, stride) in ((8, 32, 1), (20, 40, 1), (64, 64, 1), (16, 24, 2), (32, 64, 2)):  # (64 -> 64: two output blocks of the split
      # kernel, the residual prefetch of the second starts at channel 32; the last one: the stride-2 split kernel in bf16x6 mode)
      x, w = _rand((2, ci, 6, 10, 36), 91).to(DEV), _rand((co, ci, 3, 3, 3), 92, 0.1).to(DEV)
      bn = _eval_bn(co, 93)
      want = F.conv3d(x.cpu().double(), w.cpu().double(), None, stride, 1).to(DEV)
      add = _rand(tuple(want.shape), 94).to(DEV) if with_add else None
      got = HF.conv3d_bn_eval(x, w, bn, stride, add, relu)
      assert (got.double() - _unfused(bn, want, add, relu)).abs().max() < 1e-4, (ci, co, stride)
    for (cin, cout) in ((24, 40), (64, 32), (64, 64), (12, 40)):  # two output tiles (fp32 kernel), hourglass conv6 and conv5 (in bf16x6 mode the
      # split kernel's own epilogue, mode_deconv3d_fwd_split_bn), channels off the split kernel's grid
      x, w = _rand((2, cin, 3, 5, 34), 95).to(DEV), _rand((cin, cout, 3, 3, 3), 96, 0.1).to(DEV)
      bn = _eval_bn(cout, 97)
      want = F.conv_transpose3d(x.cpu().double(), w.cpu().double(), None, 2, 1, 1).to(DEV)
      add = _rand(tuple(want.shape), 98).to(DEV) if with_add else None
      assert (HF.deconv3d_bn_eval(x, w, bn, add, relu).double() - _unfused(bn, want, add, relu)).abs().max() < 1e-4, (cin, cout)


def _at_the_end_of_its_own_mapping(t):
  """A copy of t whose last byte is the last byte of a fresh device allocation: torch's caching allocator gives a request of >= 10 MB a
  segment of its own (rounded to 2 MB), so a kernel that reads even one element past the tensor leaves the mapping -- inside a cached
  block the same read lands in someone else's memory and nobody notices (round 6: a read seven planes past the last sample in the fp32
  transposed kernel survived five rounds that way and took the whole suite down when the test order changed)."""
  torch.cuda.empty_cache()
  nbytes = t.numel() * 4
  seg = 12 * 2**20
  assert nbytes <= seg and nbytes % 4 == 0
  raw = torch.empty(seg, dtype=torch.uint8, device=t.device)
  view = raw[seg - nbytes:].view(torch.float32).view(t.shape)
  view.copy_(t)
  return view, raw


def test_channel_tails_do_not_read_past_the_tensor(arith):
  """Layers whose input channels do not fill the kernels' 8 / 16 / 32-channel chunks, with the input placed at the very end of a device
  mapping: the chunk's missing channels must not be addressed at all.  (The values are checked by the other tests; here the kernels
  only have to finish.)"""
  with torch.no_grad():
    for (cin, cout) in ((12, 40), (24, 40), (20, 8)):  # transposed, fp32 kernel (+ folded BatchNorm epilogue)
      x, keep = _at_the_end_of_its_own_mapping(_rand((2, cin, 3, 5, 34), 95).to(DEV))
      w = _rand((cin, cout, 3, 3, 3), 96, 0.1).to(DEV)
      a = HF.deconv3d_bn_eval(x, w, _eval_bn(cout, 97), None, True)
      b = HF.deconv3d_fwd(x, w)
      assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    for (ci, co, stride) in ((20, 40, 1), (12, 24, 2), (36, 1, 1)):  # stride 1 / 2 and the single-channel head, fp32 kernels
      x, keep = _at_the_end_of_its_own_mapping(_rand((2, ci, 6, 10, 36), 91).to(DEV))
      w = _rand((co, ci, 3, 3, 3), 92, 0.1).to(DEV)
      y = HF.conv3d_fwd(x, w, stride)
      gw = HF.conv3d_bwd_weight(torch.ones_like(y), x, stride)
      assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(gw).all())
    for (ci, co) in ((12, 32), (20, 24)):  # 3 x 3 layers off the 16-channel grid (the fusion network's input layers)
      x, keep = _at_the_end_of_its_own_mapping(_rand((2, ci, 20, 36), 81).to(DEV))
      w = _rand((co, ci, 3, 3), 82, 0.1).to(DEV)
      y = HF.conv2d_fwd(x, w, 1)
      gw = HF.conv2d_bwd_weight(torch.ones_like(y), x, 1)
      assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(gw).all())
  torch.cuda.synchronize()


def test_conv3d_input_gradient_with_a_gradient_already_there(arith):
  """conv3d_bwd_data(..., acc=g): the sum of the input gradient and g, added in the store of the split kernels
  (mode_conv3d_bwd_data_split_acc; stride 1: the residual epilogue of conv3d_split_kernel, stride 2: of deconv3d_split_kernel, both
  with zero shifts) -- bit for bit what a separate add gives; shapes the accumulate form does not take (channels off the 32-tile,
  odd volumes, fp32-MFMA mode) go through that separate add."""
  for (ci, co, stride, shape) in ((32, 32, 1, (2, 6, 10, 36)), (64, 64, 1, (1, 4, 9, 33)), (32, 64, 2, (2, 6, 10, 36)), (64, 64, 2, (1, 4, 8, 34)),
                                  (24, 40, 1, (1, 4, 6, 34)), (40, 24, 2, (1, 4, 6, 34))):
    B, D, H, W = shape
    w = _rand((co, ci, 3, 3, 3), 401, 0.1).to(DEV)
    do, ho, wo = ((D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1)
    gy = _rand((B, co, do, ho, wo), 402).to(DEV)
    acc = _rand((B, ci, D, H, W), 403).to(DEV)
    plain = HF.conv3d_bwd_data(gy, w, (B, ci, D, H, W), stride)
    keep = acc.clone()
    got = HF.conv3d_bwd_data(gy, w, (B, ci, D, H, W), stride, acc=acc)
    assert torch.equal(got, plain + acc), (ci, co, stride, float((got - (plain + acc)).abs().max()))
    assert torch.equal(acc, keep) and got.data_ptr() != acc.data_ptr()
  for (ci, co, dil, shape) in ((64, 64, 1, (2, 20, 36)), (32, 64, 2, (1, 9, 33)), (128, 128, 1, (1, 17, 40)), (24, 40, 1, (1, 8, 34))):
    B, H, W = shape  # (mode_conv2d_bwd_data_split_acc; the last one: channels off the split kernel's grid)
    w = _rand((co, ci, 3, 3), 411, 0.1).to(DEV)
    gy = _rand((B, co, H, W), 412).to(DEV)
    acc = _rand((B, ci, H, W), 413).to(DEV)
    plain = HF.conv2d_bwd_data(gy, w, dil)
    got = HF.conv2d_bwd_data(gy, w, dil, acc=acc)
    assert torch.equal(got, plain + acc), (ci, co, dil, float((got - (plain + acc)).abs().max()))


@pytest.mark.parametrize('relu', [True, False])
def test_folded_epilogues_propagate_nan_like_torch(relu, arith):
  """A NaN activation stays NaN through conv + folded BatchNorm (+ ReLU) in BOTH arithmetics -- torch.relu(NaN) is NaN, and the
  reference's eval forward would show a diverged layer; fmaxf(NaN, 0) = 0 (or fmaxf(NaN, -inf) = -inf without ReLU) would hide it."""
  with torch.no_grad():
    x, w = _rand((1, 32, 4, 8, 32), 301).to(DEV), _rand((32, 32, 3, 3, 3), 302, 0.1).to(DEV)
    x[0, 3, 2, 4, 7] = float('nan')
    got = HF.conv3d_bn_eval(x, w, _eval_bn(32, 303), 1, None, relu)
    near = got[0, :, 1:4, 3:6, 6:9]
    assert bool(torch.isnan(near).all()), 'every output within one tap of the NaN input is NaN'
    assert int(torch.isnan(got).sum()) == near.numel() and not bool(torch.isinf(got).any())
    x2, w2 = _rand((1, 64, 16, 32), 304).to(DEV), _rand((64, 64, 3, 3), 305, 0.1).to(DEV)
    x2[0, 5, 8, 9] = float('nan')
    got = HF.conv2d_bn_eval(x2, w2, _eval_bn(64, 306), 1, None, relu)
    assert bool(torch.isnan(got[0, :, 7:10, 8:11]).all()) and int(torch.isnan(got).sum()) == 64 * 9 and not bool(torch.isinf(got).any())
  y = _rand((2, 8, 64), 307).to(DEV)
  y[1, 2, 5] = float('nan')
  bn = __import__('torch').nn.BatchNorm1d(8).to(DEV).train()
  out = HF.bn_act(bn, y, None, relu)
  assert bool(torch.isnan(out[:, 2]).all()) and int(torch.isnan(out).sum()) == 2 * 64  # that channel's statistics are NaN, like torch


@pytest.mark.parametrize('relu,with_add', FOLD_VARIANTS)
@pytest.mark.parametrize('dil', [1, 2])
def test_folded_batchnorm_conv2d_3x3(dil, relu, with_add, arith):
  import torch.nn.functional as F
  with torch.no_grad():
    x, w = _rand((2, 20, 9, 40), 99).to(DEV), _rand((40, 20, 3, 3), 100, 0.1).to(DEV)
    bn = _eval_bn(40, 101, 2)
    want = F.conv2d(x.cpu().double(), w.cpu().double(), None, 1, dil, dil).to(DEV)
    add = _rand(tuple(want.shape), 102).to(DEV) if with_add else None
    assert (HF.conv2d_bn_eval(x, w, bn, dil, add, relu).double() - _unfused(bn, want, add, relu)).abs().max() < 1e-4


@pytest.mark.parametrize('relu,with_add', FOLD_VARIANTS)
@pytest.mark.parametrize('ci,co,k,s,p', [(16, 24, 1, 1, 0), (16, 24, 3, 2, 1), (8, 8, 1, 2, 0), (3, 32, 7, 2, 3), (6, 40, 5, 1, 2), (16, 24, 1, 1, 1), (5, 7, 1, 1, 0)])
def test_folded_batchnorm_other_conv2d_layers(ci, co, k, s, p, relu, with_add):
  """Sequential(Conv2d, BatchNorm2d).eval() through stage3d.conv_bn: the 1x1 GEMM kernels, the 7x7 stem and the integer-table
  gather kernels with the folded-BatchNorm epilogue."""
  import torch.nn as nn
  import torch.nn.functional as F
  from models import stage3d
  with torch.no_grad():
    conv = nn.Conv2d(ci, co, k, s, p, bias=False).to(DEV)
    x = _rand((2, ci, 16, 24), 103).to(DEV)
    bn = _eval_bn(co, 104, 2)
    want = F.conv2d(x.cpu().double(), conv.weight.detach().cpu().double(), None, s, p).to(DEV)
    add = _rand(tuple(want.shape), 105).to(DEV) if with_add else None
    seq = nn.Sequential(conv, bn).eval()
    got = stage3d.conv_bn(seq, x, relu, add)
    assert tuple(got.shape) == tuple(want.shape)
    assert (got.double() - _unfused(bn, want, add, relu)).abs().max() < 1e-4


@pytest.mark.parametrize('relu,with_add', FOLD_VARIANTS)
@pytest.mark.parametrize('ih,iw,B', [(16, 32, 2), (64, 128, 8)])
def test_folded_batchnorm_sphere_conv(ih, iw, B, relu, with_add):
  """SphereConv.forward_bn: the general kernel (small geometry), the windowed kernel behind an NCHW caller and on plane-transposed
  storage."""
  from models.basic import SphereConv
  from models.basic.spherical_conv import sphere_conv as sc
  with torch.no_grad():
    m = SphereConv(iw, ih, 'Cassini', 16, 32, 3, 1, 1, 1, 1, False).to(DEV)
    H, W = m.position.shape[2:]
    x = _rand((B, 16, H, W), 106).to(DEV)
    bn = _eval_bn(32, 107, 2)
    want = sphere_conv_ref.forward(x.cpu().double(), m.position, m.weight.detach().cpu().double(), (1, 1), (1, 1), (1, 1), 1).to(DEV)
    add = _rand(tuple(want.shape), 108).to(DEV) if with_add else None
    ref = _unfused(bn, want, add, relu)
    assert (m.forward_bn(x, bn, add, relu).double() - ref).abs().max() < 1e-4
    if m.supports_transposed_io(B, x.device):
      with sc.transposed_io():
        got_t = m.forward_bn(HF.transpose_planes(x), bn, HF.transpose_planes(add) if add is not None else None, relu)
      assert (HF.transpose_planes(got_t).double() - ref).abs().max() < 1e-4
    else:
      assert (ih, iw) == (16, 32)


@pytest.mark.parametrize('relu', [True, False])
def test_folded_batchnorm_cost_conv(relu):
  import torch.nn.functional as F
  with torch.no_grad():
    fr, ft = _rand((2, 8, 6, 20), 109).to(DEV), _rand((2, 8, 6, 20), 110).to(DEV)
    w = _rand((12, 16, 3, 3, 3), 111, 0.1).to(DEV)
    bn = _eval_bn(12, 112)
    want = F.conv3d(mode_ref.cost_volume(fr.cpu().double(), ft.cpu().double(), 5), w.cpu().double(), None, 1, 1).to(DEV)
    assert (HF.cost_conv_bn_eval(fr, ft, w, 5, bn, relu).double() - _unfused(bn, want, None, relu)).abs().max() < 1e-4


def test_eval_forward_launches_no_batchnorm_kernel():
  """configs[1] (forward only): with the BatchNorm folded into the convolutions an eval forward of ModeDisparity records no
  bn_* region at all, and it equals the unfolded composition (fold switched off) to fp32 round-off."""
  import models
  from mode_hip import profiling
  net = models.ModeDisparity(32, 'Sphere', 128, 64, 'Cassini').to(DEV)
  net.load_state_dict(recipe.recipe_state_wc(recipe.load_manifest(), 77))  # well conditioned: round-off stays round-off
  left, right = [t.to(DEV) for t in recipe.recipe_images(2, 128, 64, 78)]
  bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
  for m in bns:
    m.momentum = 1.0  # running statistics := this batch's statistics, so that eval mode is as well conditioned as train mode
  net.train()
  with torch.no_grad():
    net(left, right)
  for m in bns:
    m.momentum = 0.1
  net.eval()
  profiling.enable(True)
  with torch.no_grad():
    folded = net(left, right)
  torch.cuda.synchronize()
  names = list(profiling.summary())
  profiling.enable(False)
  assert names and not [n for n in names if n.startswith('bn_')], names
  saved = HF.bn_foldable
  HF.bn_foldable = lambda bn, y_like=None: False
  try:
    with torch.no_grad():
      plain = net(left, right)
  finally:
    HF.bn_foldable = saved
  assert (folded - plain).abs().max() < 2e-4, float((folded - plain).abs().max())


# ------------------------------------------------------------------ 1x1 Conv2d layers: plain MFMA GEMMs (a3)
@pytest.mark.parametrize('B,Ci,Co,H,W,s', [
    (2, 32, 64, 16, 24, 1),     # layer1.0.downsample (submodule.py:167-174)
    (2, 64, 64, 16, 32, 2),     # layer2.0.downsample, stride 2
    (2, 64, 128, 12, 8, 1),     # layer4.0.downsample
    (1, 256, 128, 8, 16, 1),    # lastconv[0] (:162)
    (1, 128, 32, 8, 16, 1),     # lastconv[4]
    (1, 5, 7, 3, 12, 1),        # channel counts off the 8 / 32 blocks, pixel count off the 32-pixel segment
    (2, 12, 200, 6, 8, 1),      # more than 128 output channels (two launch rows)
    (1, 10, 6, 6, 16, 2),
    (2, 16, 24, 7, 16, 2),      # stride 2 on an ODD input height: forward, input gradient and weight gradient all take it
])
def test_conv1x1_kernels(B, Ci, Co, H, W, s):
  """mode_conv1x1_fwd / _bwd_data / _bwd_weight against torch's fp64 conv2d autograd, through the autograd Function."""
  import torch.nn as nn
  import torch.nn.functional as F
  conv = nn.Conv2d(Ci, Co, 1, s, 0, bias=False)
  with torch.no_grad():
    conv.weight.copy_(_rand((Co, Ci, 1, 1), 121, (2.0 / Ci)**0.5))
  x = _rand((B, Ci, H, W), 122)
  xa, wa = x.double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
  y_ref = F.conv2d(xa, wa, None, s)
  gy = _rand(tuple(y_ref.shape), 123)
  y_ref.backward(gy.double())
  conv = conv.to(DEV)
  xd = x.to(DEV).requires_grad_(True)
  assert HF.conv1x1_supported(xd, conv)
  y = HF.conv1x1(xd, conv)
  y.backward(gy.to(DEV))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 2e-6 * Ci * max(1.0, float(y_ref.abs().max()))
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-6 * Co * max(1.0, float(xa.grad.abs().max()))
  assert (conv.weight.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))
  # deterministic, and `accumulate` adds
  g1 = conv.weight.grad.clone()
  conv.weight.grad = None
  xd2 = x.to(DEV).requires_grad_(True)
  HF.conv1x1(xd2, conv).backward(gy.to(DEV))
  assert torch.equal(conv.weight.grad, g1) and torch.equal(xd2.grad, xd.grad)
  from models import stage3d
  with torch.no_grad():
    assert torch.equal(stage3d.conv3(conv, x.to(DEV)), y.detach())
    # eval-mode fold
    bn = _eval_bn(Co, 124, 2)
    add = _rand(tuple(y_ref.shape), 125).to(DEV)
    got = stage3d.conv_bn(nn.Sequential(conv, bn).eval(), x.to(DEV), True, add)
    assert (got.double() - _unfused(bn, y_ref.detach().to(DEV), add, True)).abs().max() < 1e-4


@pytest.mark.parametrize('B,Co,H,W', [(2, 32, 64, 32), (1, 32, 32, 128), (2, 20, 26, 70), (1, 32, 8, 8)])
def test_conv_stem_kernels(B, Co, H, W):
  """firstconv[0] = Conv2d(3, 32, 7, stride 2, padding 3) (submodule.py:155) on csrc/conv_stem.hip: forward and weight gradient
  against torch's fp64 conv2d autograd (ragged tiles, fewer output channels, images smaller than a tile); eval-mode fold."""
  import torch.nn as nn
  import torch.nn.functional as F
  from models import stage3d
  conv = nn.Conv2d(3, Co, 7, 2, 3, bias=False)
  with torch.no_grad():
    conv.weight.copy_(_rand((Co, 3, 7, 7), 131, (2.0 / 147)**0.5))
  x = _rand((B, 3, H, W), 132)
  wa = conv.weight.detach().double().requires_grad_(True)
  y_ref = F.conv2d(x.double(), wa, None, 2, 3)
  gy = _rand(tuple(y_ref.shape), 133)
  y_ref.backward(gy.double())
  conv = conv.to(DEV)
  xd = x.to(DEV)
  assert HF.conv_stem_supported(xd, conv)
  y = stage3d.conv3(conv, xd)
  assert tuple(y.shape) == tuple(y_ref.shape)
  y.backward(gy.to(DEV))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 2e-6 * 147 * max(1.0, float(y_ref.abs().max()))
  assert (conv.weight.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))
  g1 = conv.weight.grad.clone()
  conv.weight.grad = None
  stage3d.conv3(conv, xd).backward(gy.to(DEV))
  assert torch.equal(conv.weight.grad, g1)  # deterministic
  # an input that needs a gradient is not this kernel's business (integer-table kernels)
  assert not HF.conv_stem_supported(x.to(DEV).requires_grad_(True), conv)
  with torch.no_grad():
    bn = _eval_bn(Co, 134, 2)
    got = stage3d.conv_bn(nn.Sequential(conv, bn).eval(), xd, True, None)
    assert (got.double() - _unfused(bn, y_ref.detach().to(DEV), None, True)).abs().max() < 1e-4


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 64, 64, 16, 32), (1, 20, 40, 10, 36), (2, 8, 8, 2, 4)])
def test_conv2d_3x3_stride2_layer(B, Ci, Co, H, W):
  """layer2[0].conv1 (submodule.py:158): forward on the integer-table kernel, gradients = stride-1 gradients of the zero-inserted
  output gradient on the MFMA kernels (functional.Conv2d3x3S2Function), against torch's fp64 conv2d autograd."""
  import torch.nn as nn
  import torch.nn.functional as F
  from models import stage3d
  conv = nn.Conv2d(Ci, Co, 3, 2, 1, bias=False)
  with torch.no_grad():
    conv.weight.copy_(_rand((Co, Ci, 3, 3), 141, (2.0 / (9 * Ci))**0.5))
  x = _rand((B, Ci, H, W), 142)
  xa, wa = x.double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
  y_ref = F.conv2d(xa, wa, None, 2, 1)
  gy = _rand(tuple(y_ref.shape), 143)
  y_ref.backward(gy.double())
  conv = conv.to(DEV)
  xd = x.to(DEV).requires_grad_(True)
  assert HF.conv2d_3x3_s2_supported(xd, conv)
  y = stage3d.conv3(conv, xd)
  assert y.grad_fn is not None and 'Conv2d3x3S2' in type(y.grad_fn).__name__
  y.backward(gy.to(DEV))
  assert (y.detach().cpu().double() - y_ref.detach()).abs().max() < 2e-6 * 9 * Ci * max(1.0, float(y_ref.abs().max()))
  assert (xd.grad.cpu().double() - xa.grad).abs().max() < 2e-6 * 9 * Co * max(1.0, float(xa.grad.abs().max()))
  assert (conv.weight.grad.cpu().double() - wa.grad).abs().max() < 2e-5 * max(1.0, float(wa.grad.abs().max()))
  up = torch.empty(B, Co, H, W, device=DEV)
  mode_hip.check(mode_hip.lib().mode_zero_insert2(mode_hip.ptr(gy.to(DEV)), mode_hip.ptr(up), B * Co, H // 2, W // 2, None), 'mode_zero_insert2')
  want = torch.zeros(B, Co, H, W)
  want[:, :, ::2, ::2] = gy
  assert torch.equal(up.cpu(), want)


def test_batchnorm_statistics_do_not_cancel_when_mean_dominates():
  """a channel with |mean| >> std (mean 100, std 0.1; and -3000, 0.5).  E[x^2] - mean^2 in fp32 loses every digit of
  such a variance; the statistics kernel accumulates relative to a pivot inside the data instead.  Against torch's fp64
  BatchNorm: output, saved statistics (through the backward pass) and the running variance."""
  import torch.nn as nn
  g = torch.Generator().manual_seed(3)
  y = torch.randn(4, 3, 8, 16, 32, generator=g)
  y[:, 0] = y[:, 0] * 0.1 + 100.0
  y[:, 1] = y[:, 1] * 0.5 - 3000.0
  go = torch.randn(y.shape, generator=g)
  ref = nn.BatchNorm3d(3).double().train()
  y64 = y.double().requires_grad_(True)
  o64 = ref(y64)
  o64.backward(go.double())
  bn = nn.BatchNorm3d(3).to(DEV).train()
  yd = y.to(DEV).requires_grad_(True)
  out = HF.bn_act(bn, yd, None, False)
  out.backward(go.to(DEV))
  # the input itself is only known to ~1e-5 relative at 100 +- 0.1 (fp32 spacing 7.6e-6): the normalised value to ~1e-4
  assert (out.detach().cpu().double() - o64.detach()).abs().max() < 2e-3
  assert (bn.running_mean.cpu().double() - ref.running_mean).abs().max() < 1e-4 * 300
  assert ((bn.running_var.cpu().double() - ref.running_var) / ref.running_var).abs().max() < 1e-4
  assert (yd.grad.cpu().double() - y64.grad).abs().max() < 2e-3 * float(y64.grad.abs().max())
