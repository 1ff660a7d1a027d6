"""Fusion stage: oracle and module contract against the reference's golden vectors (CPU), the module on
the HIP BatchNorm kernels against the same vectors (GPU)."""
import json

import numpy as np
import pytest
import torch

import recipe
from oracle import fusion_ref

import models
from models import mode_fusion, stage3d


def _case(z):
  cfg = z['cfg']
  maxdepth, B, H, W, seed = float(cfg[0]), int(cfg[1]), int(cfg[2]), int(cfg[3]), int(cfg[4])
  channels = [int(c) for c in cfg[5:]]
  manifest = [(k, tuple(s)) for k, s in json.loads(str(z['manifest']))]
  rs = np.random.RandomState(seed + 1)
  depthes = [torch.from_numpy((rs.rand(B, 1, H, W) * maxdepth).astype(np.float32)) for _ in range(6)]
  confs = [torch.from_numpy(rs.rand(B, 1, H, W).astype(np.float32)) for _ in range(6)]
  rgbs = [torch.from_numpy(rs.rand(B, 3, H, W).astype(np.float32)) for _ in range(4)]
  gt = torch.from_numpy((rs.rand(B, H, W) * maxdepth * 1.1).astype(np.float32))
  return maxdepth, channels, manifest, recipe.recipe_state(manifest, seed), depthes, confs, rgbs, gt


# ----------------------------------------------------------------------------------------------------------- CPU tier
def test_state_dict_matches_reference_manifest():
  manifest = recipe.load_manifest('manifest_mode_fusion.json')
  net = mode_fusion.ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12})
  assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == manifest
  assert len(manifest) == 251 and sum(p.numel() for p in net.parameters()) == 3245249
  assert models.ModeFusion is mode_fusion.ModeFusion and callable(models.Baseline)
  # initialisation of the reference (:291-299): He-normal convolutions, BatchNorm (1, 0)
  bn = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
  assert all(float(m.weight.min()) == 1 and float(m.bias.abs().max()) == 0 for m in bn)
  w = net.feature_extraction.depth_layer1[0].conv1[0][0].weight
  assert abs(float(w.std()) - (2.0 / (9 * 32)) ** 0.5) < 0.1 * (2.0 / (9 * 32)) ** 0.5


def test_oracle_is_the_reference(golden):
  z = golden('fusion_tiny.npz')
  maxdepth, channels, manifest, sd, depthes, confs, rgbs, gt = _case(z)
  P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone()) for k, v in sd.items()}
  pred = fusion_ref.mode_fusion(P, depthes, confs, rgbs, maxdepth, True)
  assert np.abs(pred.detach().numpy() - z['train/pred']).max() < 1e-5
  loss = fusion_ref.training_loss(pred, gt, maxdepth)
  assert abs(float(loss) - float(z['train/loss'])) < 1e-5 * float(z['train/loss'])
  loss.backward()
  for n, s in zip(z['train/grad_names'], z['train/grad_abs_sum']):
    assert abs(float(P[str(n)].grad.double().abs().sum()) - s) <= 1e-3 * s + 1e-5, n  # zero-gradient biases hold round-off only
  for k in z.files:
    if k.startswith('bn/') and 'running' in k:
      assert np.abs(P[k[3:]].numpy() - z[k]).max() < 1e-5
  P2 = {k: torch.from_numpy(z['bn/' + k]) if 'bn/' + k in z.files else v.detach() for k, v in P.items()}
  with torch.no_grad():
    assert np.abs(fusion_ref.mode_fusion(P2, depthes, confs, rgbs, maxdepth, False).numpy() - z['eval/pred']).max() < 1e-5


def test_no_cpu_path():
  net = mode_fusion.ModeFusion(10.0, [8, 16, 32, 64], {'depth': 12, 'rgb': 12})
  x = [torch.zeros(1, 1, 16, 8)] * 6
  with pytest.raises(NotImplementedError):
    net(x, x, [torch.zeros(1, 3, 16, 8)] * 4)


def test_wiring_on_vendor_batchnorm(golden, monkeypatch):
  """The module tree evaluated with torch's own BatchNorm (the only native op of this stage swapped out) is the reference."""
  monkeypatch.setattr(stage3d, 'bn_act', stage3d.bn_act_torch)
  z = golden('fusion_tiny.npz')
  maxdepth, channels, manifest, sd, depthes, confs, rgbs, gt = _case(z)
  net = mode_fusion.ModeFusion(maxdepth, channels, {'depth': 12, 'rgb': 12})
  net.load_state_dict(sd)
  net.train()
  pred = net(depthes, confs, rgbs)
  assert pred.shape == z['train/pred'].shape and np.abs(pred.detach().numpy() - z['train/pred']).max() < 1e-5
  net.eval()
  with torch.no_grad():
    assert np.abs(net(depthes, confs, rgbs).numpy() - z['eval/pred']).max() < 1e-5
  base = mode_fusion.Baseline(maxdepth)
  base.eval()
  with torch.no_grad():
    assert base([d[:, :, :16, :8] for d in depthes]).shape == (2, 1, 16, 8)


# ----------------------------------------------------------------------------------------------------------- GPU tier
@pytest.mark.gpu
def test_gpu_fusion_train_and_eval(golden):
  """fp32 on the GPU against the fp64 evaluation of the same network: as close as the reference's own fp32 run (mean error
  within 2x, max within 3x, as for the disparity stage)."""
  dev = 'cuda:0'
  z = golden('fusion_tiny.npz')
  maxdepth, channels, manifest, sd, depthes, confs, rgbs, gt = _case(z)
  net = mode_fusion.ModeFusion(maxdepth, channels, {'depth': 12, 'rgb': 12}).to(dev)
  net.load_state_dict(sd)
  dd, cc, rr = [t.to(dev) for t in depthes], [t.to(dev) for t in confs], [t.to(dev) for t in rgbs]

  def check(name, got, ref32, truth):
    err, ref_err = np.abs(got - truth), np.abs(ref32 - truth)
    print('%s: |gpu-truth64| max %.2e mean %.2e; reference itself max %.2e mean %.2e' % (name, err.max(), err.mean(), ref_err.max(), ref_err.mean()))
    assert err.max() <= max(1e-4, 3 * ref_err.max()) and err.mean() <= max(1e-6, 2 * ref_err.mean())

  from mode_hip import no_vendor
  net.train()
  with no_vendor.no_vendor_arithmetic() as guard:  # the whole forward on the hand-written kernels
    pred = net(dd, cc, rr)
  assert guard.seen > 0
  check('train', pred.detach().cpu().numpy().astype(np.float64), z['train/pred'], z['truth64/train_pred'])
  loss = fusion_ref.training_loss(pred, gt.to(dev), maxdepth)
  assert abs(float(loss) - float(z['train/loss'])) < 1e-4 * float(z['train/loss'])
  loss.backward()
  grads = dict(net.named_parameters())
  for n, s in zip(z['train/grad_names'], z['train/grad_abs_sum']):
    # (the bias of a transposed convolution that feeds a BatchNorm has zero gradient: both sides hold round-off there)
    assert abs(float(grads[str(n)].grad.double().abs().sum()) - s) <= 2e-2 * s + 1e-5, n
  sd_after = net.state_dict()
  for k in z.files:
    if k.startswith('bn/'):
      a = sd_after[k[3:]].cpu().numpy()
      assert np.abs(a - z[k]).max() <= 1e-4 * max(1.0, np.abs(z[k]).max()), k
  net.eval()
  with torch.no_grad(), no_vendor.no_vendor_arithmetic():
    out = net(dd, cc, rr)
  check('eval', out.cpu().numpy().astype(np.float64), z['eval/pred'], z['truth64/eval_pred'])


@pytest.mark.gpu
def test_gpu_fusion_full_size_step():
  """The configuration of train_fusion.py:64 at 1024x512, batch 1: one training step runs and is finite."""
  dev = 'cuda:0'
  net = mode_fusion.ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12}).to(dev).train()
  g = torch.Generator(device='cpu').manual_seed(1)
  depthes = [(torch.rand(1, 1, 1024, 512, generator=g) * 50).to(dev) for _ in range(6)]
  confs = [torch.rand(1, 1, 1024, 512, generator=g).to(dev) for _ in range(6)]
  rgbs = [torch.rand(1, 3, 1024, 512, generator=g).to(dev) for _ in range(4)]
  gt = (torch.rand(1, 1024, 512, generator=g) * 60).to(dev)
  from mode_hip import no_vendor
  with no_vendor.no_vendor_arithmetic():
    pred = net(depthes, confs, rgbs)
  assert pred.shape == (1, 1, 1024, 512) and float(pred.min()) >= 0 and float(pred.max()) <= 1000
  with no_vendor.no_vendor_arithmetic() as guard:  # the backward too (the dispatch mode travels to autograd's device thread)
    fusion_ref.training_loss(pred, gt, 1000).backward()
  assert guard.seen > 0
  assert all(torch.isfinite(p.grad).all() for p in net.parameters())
  # inference at the same size (BASELINE configs[4] names the forward): BatchNorm folded, still no vendor arithmetic
  net.eval()
  with torch.no_grad(), no_vendor.no_vendor_arithmetic():
    out = net(depthes, confs, rgbs)
  assert out.shape == (1, 1, 1024, 512) and bool(torch.isfinite(out).all())


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 5, 8, 12), (1, 3, 9, 7), (2, 4, 16, 32), (1, 2, 2, 2)])
def test_gpu_maxpool2x2_is_torchs(shape):
  """mode_maxpool2x2_fwd / _bwd against torch's max_pool2d on the CPU: values, and the gradient's routing (ties go to the first element
  in scan order, NaN wins), odd sizes (floor), bit for bit."""
  from mode_hip import functional as HF
  import torch.nn.functional as F
  g = torch.Generator().manual_seed(3)
  x = torch.randn(*shape, generator=g)
  x[0, 0, :2, :2] = 1.5  # a tie
  if shape[2] >= 4:
    x[0, 1, 2, 3] = float('nan')
  xr = x.clone().requires_grad_(True)
  y_ref = F.max_pool2d(xr, 2, 2)
  go = torch.randn(*y_ref.shape, generator=g)
  y_ref.backward(go)
  xd = x.to('cuda:0').requires_grad_(True)
  y = HF.maxpool2x2(xd)
  y.backward(go.to('cuda:0'))
  assert torch.equal(y.detach().cpu().nan_to_num(nan=7.0), y_ref.detach().nan_to_num(nan=7.0))
  assert torch.equal(xd.grad.cpu(), xr.grad)
  pool = torch.nn.MaxPool2d(2, stride=2)
  assert HF.maxpool2x2_supported(xd, pool) and not HF.maxpool2x2_supported(xd, torch.nn.MaxPool2d(3, stride=2))


@pytest.mark.gpu
@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 16, 8, 6, 8), (1, 64, 32, 16, 12), (1, 256, 128, 8, 4)])
def test_gpu_deconv2x2_against_float64(B, Ci, Co, H, W):
  """ConvTranspose2d(Ci, Co, 2, 2) as the 1x1 GEMM of csrc/conv1x1.hip + the rearrangement of csrc/fusion_ops.hip: forward, input
  gradient, weight gradient, bias gradient against torch's float64 on the CPU; the eval form with the BatchNorm folded in."""
  from mode_hip import functional as HF
  torch.manual_seed(5)
  conv = torch.nn.ConvTranspose2d(Ci, Co, 2, 2)
  x = torch.randn(B, Ci, H, W)
  go = torch.randn(B, Co, 2 * H, 2 * W)
  c64 = torch.nn.ConvTranspose2d(Ci, Co, 2, 2).double()
  c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
  x64 = x.double().requires_grad_(True)
  y64 = c64(x64)
  y64.backward(go.double())
  convd = conv.to('cuda:0')
  xd = x.to('cuda:0').requires_grad_(True)
  assert HF.deconv2x2_supported(xd, convd)
  y = HF.deconv2x2(xd, convd)
  y.backward(go.to('cuda:0'))
  tol = 2e-6 * Ci ** 0.5
  assert float((y.detach().cpu().double() - y64.detach()).abs().max()) <= tol * max(1.0, float(y64.abs().max()))
  assert float((xd.grad.cpu().double() - x64.grad).abs().max()) <= 4 * tol * max(1.0, float(x64.grad.abs().max()))
  gw64, gb64 = c64.weight.grad, c64.bias.grad
  assert float((convd.weight.grad.cpu().double() - gw64).abs().max()) <= 2e-6 * (B * H * W) ** 0.5 * max(1.0, float(gw64.abs().max()))
  assert float((convd.bias.grad.cpu().double() - gb64).abs().max()) <= 1e-5 * max(1.0, float(gb64.abs().max()))
  bn = torch.nn.BatchNorm2d(Co)
  with torch.no_grad():
    bn.running_mean.normal_()
    bn.running_var.uniform_(0.5, 2.0)
    bn.weight.uniform_(0.5, 1.5)
    bn.bias.normal_()
  bn.eval()
  ref = torch.relu(bn.double()(y64.detach()))
  bn = bn.float().to('cuda:0')
  with torch.no_grad():
    got = HF.deconv2x2_bn_eval(xd.detach(), convd, bn, True)
  assert float((got.cpu().double() - ref).abs().max()) <= 4 * tol * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,H,W', [(2, 32, 8, 12), (1, 8, 6, 10), (1, 32, 64, 32)])
def test_gpu_conv1x1_sigmoid_against_float64(B, C, H, W):
  """Conv2d(C, 1, 1, bias=True) + Sigmoid in one pass (mode_conv1x1_sigmoid_fwd / _bwd) against torch's float64 on the CPU."""
  from mode_hip import functional as HF
  torch.manual_seed(6)
  conv = torch.nn.Conv2d(C, 1, 1, bias=True)
  x = torch.randn(B, C, H, W)
  go = torch.randn(B, 1, H, W)
  c64 = torch.nn.Conv2d(C, 1, 1, bias=True).double()
  c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
  x64 = x.double().requires_grad_(True)
  s64 = torch.sigmoid(c64(x64))
  s64.backward(go.double())
  convd = conv.to('cuda:0')
  xd = x.to('cuda:0').requires_grad_(True)
  assert HF.conv1x1_sigmoid_supported(xd, convd)
  s = HF.conv1x1_sigmoid(xd, convd)
  s.backward(go.to('cuda:0'))
  assert float((s.detach().cpu().double() - s64.detach()).abs().max()) <= 2e-6
  assert float((xd.grad.cpu().double() - x64.grad).abs().max()) <= 2e-6 * max(1.0, float(x64.grad.abs().max()))
  assert float((convd.weight.grad.cpu().double() - c64.weight.grad).abs().max()) <= 1e-5 * max(1.0, float(c64.weight.grad.abs().max()))
  assert float((convd.bias.grad.cpu().double() - c64.bias.grad).abs().max()) <= 1e-5 * max(1.0, float(c64.bias.grad.abs().max()))
