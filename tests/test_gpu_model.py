"""GPU (-m gpu): the drop-in ModeDisparity module on the HIP path against the golden vectors that the imported
reference produced (tests/golden/*.npz).  Tolerance from BASELINE.json's north_star: 1e-3 abs on the disparity."""
import numpy as np
import pytest
import torch

import recipe
from oracle import mode_ref

import models
import mode_hip

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DISP_TOL = 1e-3


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()


def _setup(z, bn_from_fixture=False):
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  sd = recipe.recipe_state(recipe.load_manifest(), seed)
  if bn_from_fixture:
    for k in z.files:
      if k.startswith('bn/'):
        sd[k[3:]] = torch.from_numpy(z[k]).clone()
  net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini').to(DEV)
  net.load_state_dict(sd)
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity(B, H, W, seed + 2, maxdisp)
  return net, left.to(DEV), right.to(DEV), gt.to(DEV), maxdisp


def test_tiny_train_forward_backward(golden):
  z = golden('model_tiny.npz')
  net, left, right, gt, maxdisp = _setup(z)
  net.train()
  preds = net(left, right)
  for i, p in enumerate(preds):
    assert np.abs(p.detach().cpu().numpy() - z['train/pred%d' % (i + 1)]).max() < DISP_TOL
  mask = ~torch.isnan(gt)
  loss = mode_ref.training_loss(preds, gt, mask)
  assert abs(float(loss.detach()) - float(z['train/loss'])) < 1e-4 * float(z['train/loss'])
  loss.backward()
  grads = dict(net.named_parameters())
  worst = 0.0
  for n, s, idx, val in zip(z['train/grad_names'], z['train/grad_abs_sum'], z['train/grad_idx'], z['train/grad_val']):
    g = grads[str(n)].grad.detach().cpu().reshape(-1).double()
    rel = abs(float(g.abs().sum()) - s) / (s + 1e-7)
    worst = max(worst, rel)
    assert rel < 5e-3, (str(n), rel)
    assert np.allclose(g[idx].numpy(), val, rtol=2e-2, atol=5e-3 * s / g.numel() + 1e-7), str(n)
  print('worst relative |grad| sum error', worst)


def test_tiny_eval_and_confidence(golden):
  z = golden('model_tiny.npz')
  net, left, right, gt, maxdisp = _setup(z, bn_from_fixture=True)
  net.eval()
  net.out_conf = True
  with torch.no_grad():
    pred, conf = net(left, right)
  assert np.abs(pred.cpu().numpy() - z['eval/pred3']).max() < DISP_TOL
  assert np.abs(conf.cpu().numpy() - z['eval/conf']).max() < 1e-3


def test_cfg1_eval(golden):
  """BASELINE configs[0] shape (Cassini 512x256, 64 disparities) on the GPU path."""
  z = golden('model_cfg1.npz')
  net, left, right, gt, maxdisp = _setup(z, bn_from_fixture=True)
  net.eval()
  with torch.no_grad():
    pred = net(left, right)
  assert pred.shape == (1, 1, 512, 256)
  assert np.abs(pred[:, :, ::4, ::4].cpu().numpy() - z['eval/pred3']).max() < DISP_TOL
  assert abs(float(pred.double().mean()) - float(z['eval/pred3_mean'])) < 1e-4


def test_cfg1_train_outputs_and_grads(golden):
  z = golden('model_cfg1.npz')
  net, left, right, gt, maxdisp = _setup(z)
  net.train()
  preds = net(left, right)
  for i, p in enumerate(preds):
    assert np.abs(p.detach()[:, :, ::4, ::4].cpu().numpy() - z['train/pred%d' % (i + 1)]).max() < DISP_TOL
  loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
  assert abs(float(loss.detach()) - float(z['train/loss'])) < 1e-4 * float(z['train/loss'])
  loss.backward()
  grads = dict(net.named_parameters())
  for n, s in zip(z['train/grad_names'], z['train/grad_abs_sum']):
    g = grads[str(n)].grad
    assert abs(float(g.double().abs().sum()) - s) <= 1e-2 * s + 1e-7, str(n)


def test_smoke_entry():
  import __graft_entry__
  __graft_entry__.smoke()
