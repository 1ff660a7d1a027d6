"""CPU: pin the oracle (oracle/) against the golden vectors made from the imported reference."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import recipe
from oracle import mode_ref, sphere_conv_ref

GOLDEN = os.path.dirname(os.path.abspath(recipe.__file__))


def _sha(a):
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ sampling table (a5)
@pytest.mark.parametrize('typ,ih,iw', [('ERP', 8, 16), ('Cassini', 16, 8)])
def test_position_small_bit_exact(golden, typ, ih, iw):
  g = golden('positions.npz')['%s_%dx%d' % (typ, ih, iw)]
  p = mode_ref.sphere_position(ih, iw, typ).numpy()
  assert p.shape == g.shape and p.dtype == np.float32
  assert np.array_equal(p, g)


@pytest.mark.parametrize('key', ['Cassini_256x128', 'Cassini_128x64', 'ERP_128x256', 'Cassini_16x8', 'Cassini_512x256'])
def test_position_sha(golden, key):
  with open(os.path.join(GOLDEN, 'positions_meta.json')) as f:
    meta = json.load(f)[key]
  typ, dims = key.split('_')
  ih, iw = map(int, dims.split('x'))
  p = mode_ref.sphere_position(ih, iw, typ).numpy()
  z = golden('positions.npz')
  assert list(p.shape) == meta['shape']
  assert np.array_equal(p.reshape(-1)[z[key + '_idx']], z[key + '_val'])
  assert _sha(p) == meta['sha256']
  assert not np.isnan(p).any()


# ------------------------------------------------------------------ native op restatement (a7/a8)
@pytest.mark.parametrize('name', ['erp_s1', 'cas_s1', 'erp_s2', 'cas_g2'])
def test_sphere_conv_restatement(golden, name):
  z = golden('sphere_conv.npz')
  ih, iw, ci, co, s, g = [int(v) for v in z[name + '/cfg']]
  typ = str(z[name + '/type'])
  pos = mode_ref.sphere_position(ih, iw, typ)
  x, w, gy = (torch.from_numpy(z['%s/%s' % (name, k)]) for k in ('x', 'w', 'gy'))
  cfg = ((s, s), (1, 1), (1, 1), g)
  y = sphere_conv_ref.forward(x.double(), pos, w.double(), *cfg)
  assert np.allclose(y.numpy(), z[name + '/y'], rtol=0, atol=1e-12)
  gx, gw = sphere_conv_ref.backward(x.double(), pos, w.double(), gy.double(), *cfg)
  assert np.allclose(gx.numpy(), z[name + '/gx'], rtol=0, atol=1e-11)
  assert np.allclose(gw.numpy(), z[name + '/gw'], rtol=0, atol=1e-10)
  # independent formulation + autograd
  xa, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
  y2 = sphere_conv_ref.forward_grid_sample(xa, pos, wa, *cfg)
  y2.backward(gy.double())
  assert (y2 - y).abs().max() < 1e-12
  assert (xa.grad - gx).abs().max() < 1e-11 and (wa.grad - gw).abs().max() < 1e-10
  # fp32 direct form stays within fp32 round-off of the fp64 truth
  y32 = sphere_conv_ref.forward(x, pos, w, *cfg)
  assert (y32.double() - y).abs().max() < 2e-5


def test_sphere_conv_autograd_wrapper_matches_explicit_backward():
  pos = mode_ref.sphere_position(16, 8, 'Cassini')
  g = torch.Generator().manual_seed(0)
  x = torch.randn(1, 3, 16, 8, generator=g, dtype=torch.float64, requires_grad=True)
  w = torch.randn(5, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
  y = sphere_conv_ref.sphere_conv(x, pos, w, None, 1, 1, 1, 1)
  gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
  y.backward(gy)
  gx, gw = sphere_conv_ref.backward(x.detach(), pos, w.detach(), gy, (1, 1), (1, 1), (1, 1), 1)
  assert torch.equal(gx, x.grad) and torch.equal(gw, w.grad)


# ------------------------------------------------------------------ cost volume (a9)
def test_cost_volume_matches_reference_capture(golden):
  z = golden('model_tiny.npz')
  fl, fr = torch.from_numpy(z['train/fea_left']), torch.from_numpy(z['train/fea_right'])
  cost = mode_ref.cost_volume(fl, fr, 16 // 4)
  assert np.array_equal(cost.numpy(), z['train/cost'])
  assert _sha(cost.numpy()) == str(z['train/cost_sha256'])


def test_cost_volume_edge_cases():
  fl = torch.arange(2 * 3 * 2 * 5, dtype=torch.float32).reshape(2, 3, 2, 5) + 1
  fr = -fl
  c = mode_ref.cost_volume(fl, fr, 7)  # more disparity levels than columns: fully-zero slices
  assert c.shape == (2, 6, 7, 2, 5)
  assert torch.equal(c[:, :3, 0], fl) and torch.equal(c[:, 3:, 0], fr)
  assert torch.equal(c[:, 3:, 2, :, 2:], fr[..., :3]) and (c[:, :, 2, :, :2] == 0).all()
  assert (c[:, :, 5:] == 0).all()


# ------------------------------------------------------------------ hourglass (a11)
@pytest.mark.parametrize('tag', ['none', 'both'])
def test_hourglass(golden, tag):
  z = golden('hourglass.npz')
  manifest = [(k, tuple(s)) for k, s in json.loads(str(z['manifest']))]
  P = {'hg.' + k: v for k, v in recipe.recipe_state(manifest, 11).items()}
  for k, v in P.items():
    if v.is_floating_point() and 'running' not in k:
      v.requires_grad_(True)
  x = torch.from_numpy(z['x']).requires_grad_(True)
  a = torch.from_numpy(z['presqu']) if tag == 'both' else None
  b = torch.from_numpy(z['postsqu']) if tag == 'both' else None
  out, pre, post = mode_ref.hourglass(P, 'hg', x, a, b, True)
  for name, t in (('out', out), ('pre', pre), ('post', post)):
    assert np.allclose(t.detach().numpy(), z['%s/%s' % (tag, name)], rtol=1e-5, atol=1e-5), name
  (out * torch.from_numpy(z[tag + '/gout'])).sum().backward()
  assert np.allclose(x.grad.numpy(), z[tag + '/gx'], rtol=1e-4, atol=1e-5)
  for k, shape in manifest:
    if ('grad/' + k) in [n.split('/', 1)[1] for n in z.files if n.startswith(tag + '/grad/')]:
      g = z['%s/grad/%s' % (tag, k)]
      assert np.allclose(P['hg.' + k].grad.numpy(), g, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(g).max())), k


# ------------------------------------------------------------------ whole model (a2, a3, a10-a15)
def _model_inputs(z):
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  P = recipe.recipe_state(recipe.load_manifest(), seed)
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity(B, H, W, seed + 2, maxdisp)
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
  return maxdisp, H, W, B, P, left, right, gt, pos


def _load_bn(P, z):
  for k in z.files:
    if k.startswith('bn/'):
      P[k[3:]] = torch.from_numpy(z[k]).clone()


def test_model_tiny_train_forward_backward(golden):
  z = golden('model_tiny.npz')
  maxdisp, H, W, B, P, left, right, gt, pos = _model_inputs(z)
  params = [k for k, v in P.items() if v.is_floating_point() and 'running' not in k]
  for k in params:
    P[k].requires_grad_(True)
  taps = {}
  preds = mode_ref.mode_disparity(P, left, right, maxdisp, pos, True, taps=taps)
  assert np.allclose(taps['fea_left'].detach().numpy(), z['train/fea_left'], rtol=1e-4, atol=1e-4)
  assert np.array_equal(taps['cost'].detach().numpy() != 0, z['train/cost'] != 0)
  for i, p in enumerate(preds):
    assert np.abs(p.detach().numpy() - z['train/pred%d' % (i + 1)]).max() < 1e-3
  mask = ~torch.isnan(gt)
  loss = mode_ref.training_loss(preds, gt, mask)
  assert abs(float(loss.detach()) - float(z['train/loss'])) < 1e-4 * float(z['train/loss'])
  loss.backward()
  names = [str(n) for n in z['train/grad_names']]
  assert names == params
  for n, s, idx, val in zip(names, z['train/grad_abs_sum'], z['train/grad_idx'], z['train/grad_val']):
    g = P[n].grad.reshape(-1).double()
    assert abs(float(g.abs().sum()) - s) <= 2e-3 * s + 1e-7, n
    assert np.allclose(g[idx].numpy(), val, rtol=5e-3, atol=2e-3 * s / g.numel() + 1e-8), n


def test_model_tiny_eval_and_confidence(golden):
  z = golden('model_tiny.npz')
  maxdisp, H, W, B, P, left, right, gt, pos = _model_inputs(z)
  _load_bn(P, z)
  taps = {}
  with torch.no_grad():
    pred, conf = mode_ref.mode_disparity(P, left, right, maxdisp, pos, False, out_conf=True, taps=taps)
  assert np.abs(taps['classif3_raw'].numpy() - z['eval/logits3']).max() < 2e-4  # classif3 module output
  assert np.abs(pred.numpy() - z['eval/pred3']).max() < 1e-3
  assert np.abs(conf.numpy() - z['eval/conf']).max() < 1e-4


def test_model_cfg1_eval(golden):
  """BASELINE.json configs[0]: 256x512 ERP (Cassini 512x256), 64 disparities, CPU."""
  z = golden('model_cfg1.npz')
  maxdisp, H, W, B, P, left, right, gt, pos = _model_inputs(z)
  assert (maxdisp, H, W) == (64, 512, 256)
  _load_bn(P, z)
  with torch.no_grad():
    pred = mode_ref.mode_disparity(P, left, right, maxdisp, pos, False)
  assert np.abs(pred[:, :, ::4, ::4].numpy() - z['eval/pred3']).max() < 1e-3
  assert abs(float(pred.double().mean()) - float(z['eval/pred3_mean'])) < 1e-4


# ------------------------------------------------------------------ well-conditioned fixtures (the 1e-3 bar itself)
@pytest.mark.parametrize('tag', ['wc_tiny', 'wc_cfg1', 'peaked_tiny'])
def test_model_wellconditioned_matches_the_reference_to_1e4(golden, tag):
  """The oracle on the fixtures tests/test_gpu_parity.py holds the HIP path to (made by the imported reference,
  tests/golden/make_golden_wc.py): same torch CPU kernels in the same order, so the agreement is at round-off level --
  outputs to 1e-4 px (a tenth of the north_star's bound), gradients to the reference's own fp32 reproducibility."""
  z = golden('model_%s.npz' % tag)
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  sub = int(z['sub'])
  assert float(z['truth64/train_E_ref']) <= 1.5e-4 and float(z['truth64/eval_E_ref']) <= 1.5e-4  # what "well conditioned" means
  if tag.startswith('peaked'):  # ... and what "peaked" means: the softmax of the trained heads holds most of its mass within +-1 px
    assert float(z['eval/conf'].mean()) >= 0.5 and float(z['eval/conf'].mean()) >= 3.0 * 3.0 / maxdisp
  P = recipe.fixture_state(z)
  left, right, gt = recipe.fixture_inputs(z)
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
  params = [k for k, v in P.items() if v.is_floating_point() and 'running' not in k]
  for k in params:
    P[k].requires_grad_(True)
  preds = mode_ref.mode_disparity(P, left, right, maxdisp, pos, True)
  for i, p in enumerate(preds):
    assert np.abs(p.detach()[:, :, ::sub, ::sub].numpy() - z['train/pred%d' % (i + 1)]).max() <= 1e-4
    assert np.abs(torch.nn.functional.avg_pool2d(p.detach().double(), 8).numpy() - z['train/pred%d_block' % (i + 1)]).max() <= 1e-4
  loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
  assert abs(float(loss.detach()) - float(z['train/loss'])) <= 1e-5 * float(z['train/loss'])
  loss.backward()
  assert [str(n) for n in z['train/grad_names']] == params
  for i, n in enumerate(params):
    g = P[n].grad.reshape(-1).double().numpy()
    proj = recipe.projection_signs(seed, i, g.size, z['train/grad_proj'].shape[1]).astype(np.float64) @ g
    rel = float(np.sqrt(np.mean((proj - z['train/grad_proj'][i])**2))) / (float(z['train/grad_norm'][i]) + 1e-300)
    assert rel <= max(1e-3, 5.0 * float(z['truth64/grad_rel_l2'][i])), (n, rel)
  for k in z.files:
    if k.startswith('bn/'):
      P[k[3:]] = torch.from_numpy(z[k]).clone()
  with torch.no_grad():
    pred, conf = mode_ref.mode_disparity({k: v.detach() for k, v in P.items()}, left, right, maxdisp, pos, False, out_conf=True)
  assert np.abs(pred[:, :, ::sub, ::sub].numpy() - z['eval/pred3']).max() <= 1e-4
  assert np.abs(conf[:, :, ::sub, ::sub].numpy() - z['eval/conf']).max() <= 1e-4


def test_wellconditioned_full_size_fixture_is_well_conditioned(golden):
  """The benchmark-size fixture (1024 x 512, 192 disparities) is only evaluated on the GPU tier; here: its header."""
  z = golden('model_wc_full.npz')
  assert [int(v) for v in z['cfg'][:4]] == [192, 1024, 512, 1]
  assert float(z['truth64/train_E_ref']) <= 1.5e-4 and float(z['truth64/eval_E_ref']) <= 1.5e-4
  assert z['train/pred3'].shape == (1, 1, 128, 64) and z['train/pred3_block'].shape == (1, 1, 128, 64)
  assert len(z['train/grad_names']) == 243


# ------------------------------------------------------------------ tap-wise fp64 conv oracle (full-size GPU tests)
@pytest.mark.parametrize('B,Ci,Co,D,H,W,stride', [(2, 3, 5, 4, 6, 8, 1), (1, 4, 2, 6, 4, 10, 2), (1, 2, 3, 1, 1, 2, 1), (2, 5, 4, 8, 2, 6, 2)])
def test_tapwise_conv_oracle_equals_torch_conv3d(B, Ci, Co, D, H, W, stride):
  """oracle/conv_ref.py (27 GEMMs on strided views) against torch's CPU conv3d / conv_transpose3d and their autograd -- the
  kernels oracle/mode_ref.py and the imported reference itself compute with."""
  import torch.nn.functional as F
  from oracle import conv_ref
  g = torch.Generator().manual_seed(7)
  x = torch.randn(B, Ci, D, H, W, generator=g, dtype=torch.float64, requires_grad=True)
  w = torch.randn(Co, Ci, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
  y = F.conv3d(x, w, None, stride, 1)
  gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
  y.backward(gy)
  assert (conv_ref.conv3d_fwd(x.detach(), w.detach(), stride) - y.detach()).abs().max() < 1e-12
  assert (conv_ref.conv3d_bwd_weight(gy, x.detach(), stride) - w.grad).abs().max() < 1e-11
  assert (conv_ref.conv3d_bwd_data(gy, w.detach(), x.shape, stride) - x.grad).abs().max() < 1e-12
  wt = torch.randn(Ci, Co, 3, 3, 3, generator=g, dtype=torch.float64)
  assert (conv_ref.deconv3d_fwd(x.detach(), wt) - F.conv_transpose3d(x.detach(), wt, None, 2, 1, 1)).abs().max() < 1e-12


def test_three_training_steps_of_the_oracle_follow_the_imported_reference(golden):
  """model_steps_tiny.npz (tests/golden/make_golden_steps.py): three iterations of the reference's own loop body (train_disparity.py:147-161,
  Adam lr 1e-3).  The CPU restatement, stepped by the same optimizer, reproduces the losses, the BatchNorm running statistics after step 3
  and the parameter update -- the fixture the GPU tier (tests/test_gpu_steps.py) then holds the product's graph-replayed step to."""
  z = golden('model_steps_tiny.npz')
  maxdisp, H, W, B, seed, K = [int(v) for v in z['cfg']]
  P = recipe.recipe_state_wc(recipe.load_manifest(), seed)
  names = [str(n) for n in z['names']]
  for k in names:
    P[k].requires_grad_(True)
  p0 = {k: P[k].detach().clone().double() for k in names}
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity_smooth(B, H, W, seed + 2, maxdisp)
  mask = ~torch.isnan(gt)
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
  opt = torch.optim.Adam([P[k] for k in names], lr=1e-3, betas=(0.9, 0.999))
  for step in range(K):
    opt.zero_grad()
    loss = mode_ref.training_loss(mode_ref.mode_disparity(P, left, right, maxdisp, pos, True), gt, mask)
    loss.backward()
    opt.step()
    assert abs(float(loss.detach()) - z['loss'][step]) <= (2e-5 if step == 0 else 2e-4) * z['loss'][step], (step, float(loss.detach()), z['loss'][step])
  for key in z.files:
    if key.startswith('bn/'):
      want = z[key].astype(np.float64)
      assert np.abs(P[key[3:]].double().numpy() - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), key
  rel = []
  for i, k in enumerate(names):
    d = (P[k].detach().double() - p0[k]).reshape(-1).numpy()
    proj = recipe.projection_signs(seed, i, d.size, z['dproj'].shape[1]).astype(np.float64) @ d
    rel.append((float(((proj - z['dproj'][i])**2).mean()) / max(z['dnorm'][i]**2, 1e-30))**0.5)
  rel = np.array(rel)
  print('relative L2 of the parameter update per tensor: median %.3e, max %.3e' % (np.median(rel), rel.max()))
  assert np.median(rel) <= 2e-2 and rel.max() <= 0.5
