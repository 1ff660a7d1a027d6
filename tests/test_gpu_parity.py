"""GPU (-m gpu): the north_star's parity bar, asserted as stated.

    "Outputs match the reference PyTorch-CPU ModeDisparity on identical inputs within 1e-3 abs on the final disparity map."

Fixtures: tests/golden/model_wc_{tiny,cfg1,full}.npz -- outputs of the IMPORTED reference (tests/golden/make_golden_wc.py) on the
well-conditioned recipe state (recipe.recipe_state_wc), where the reference's own fp32 run is reproducible to E_ref ~ 1e-5 .. 1e-4 px
(stored in the fixture), at three sizes: 64x32 / 16 disparities (B=2), BASELINE configs[0] = 512x256 / 64 (B=1) and the benchmark
size 1024x512 / 192 (B=1, configs[1] forward and the per-sample share of configs[2]).  Asserted, train mode (pred1-3) and eval mode
(pred3):
  * max |HIP - reference fp32| <= 1e-3 px on every stored pixel (all pixels at tiny, every 4th / 8th at the larger sizes), and
    on the 8x8 block means of ALL pixels;
  * loss to 2e-5 relative;
  * parameter gradients: relative L2 error per tensor <= max(1e-3, 5 x the reference's own fp32-vs-fp64 error of that tensor),
    estimated from 16 Rademacher projections stored in the fixture (E <e, v>^2 = |e|^2), plus 32 sampled entries per tensor
    (median at the L2 level, maximum within 100x of it).
    (The extractor's gradients are a ~1e-3 residual after ~60 BatchNorm backward passes: the reference's own fp32 gradients
    there are only good to 1e-3 .. 4e-3 relative, so 1e-3 cannot be asked of them; the 3-D stage is held to 1e-3.)
The ill-conditioned random-init fixtures of round 1 (tests/test_gpu_model.py) stay as the stress tier."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe
from oracle import mode_ref

import models
import mode_hip

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DISP_TOL = 1e-3  # BASELINE.json north_star


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()


def _load(z, bn_from_fixture=False):
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  mix, logit_scale = [float(v) for v in z['wc']]
  sd = recipe.recipe_state_wc(recipe.load_manifest(), seed, mix, logit_scale)
  if bn_from_fixture:
    for k in z.files:
      if k.startswith('bn/'):
        sd[k[3:]] = torch.from_numpy(z[k]).clone()
  net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini').to(DEV)
  net.load_state_dict(sd)
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity_smooth(B, H, W, seed + 2, maxdisp)
  return net, left.to(DEV), right.to(DEV), gt.to(DEV), seed


def _check_pred(name, got, z, key, e_ref):
  sub = int(z['sub'])
  g = got.detach()
  d_pix = np.abs(g[:, :, ::sub, ::sub].cpu().numpy().astype(np.float64) - z[key])
  d_blk = np.abs(F.avg_pool2d(g.double(), 8).cpu().numpy() - z[key + '_block'])
  print('%s: max|HIP - reference fp32| = %.3e px (stored pixels), %.3e (8x8 block means, all pixels); reference vs fp64: %.3e' %
        (name, d_pix.max(), d_blk.max(), float(e_ref)))
  assert d_pix.max() <= DISP_TOL, (name, d_pix.max())
  assert d_blk.max() <= DISP_TOL, (name, d_blk.max())


@pytest.mark.parametrize('tag', ['tiny', 'cfg1', 'full'])
def test_train_outputs_and_gradients_within_1e3_of_the_reference(golden, tag):
  z = golden('model_wc_%s.npz' % tag)
  net, left, right, gt, seed = _load(z)
  net.train()
  preds = net(left, right)
  for i, p in enumerate(preds):
    _check_pred('%s train pred%d' % (tag, i + 1), p, z, 'train/pred%d' % (i + 1), z['truth64/train_E_ref'])
  loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
  ref_loss = float(z['train/loss'])
  assert abs(float(loss.detach()) - ref_loss) <= 2e-5 * ref_loss, (float(loss.detach()), ref_loss)
  loss.backward()
  grads = dict(net.named_parameters())
  own = z['truth64/grad_rel_l2'] if 'truth64/grad_rel_l2' in z.files else None
  worst, worst_3d = 0.0, 0.0
  for i, name in enumerate(z['train/grad_names']):
    name = str(name)
    g = grads[name].grad.detach().cpu().reshape(-1).double().numpy()
    norm = float(z['train/grad_norm'][i])
    proj = recipe.projection_signs(seed, i, g.size, z['train/grad_proj'].shape[1]).astype(np.float64) @ g
    rel = float(np.sqrt(np.mean((proj - z['train/grad_proj'][i])**2))) / (norm + 1e-300)
    # the reference's own fp32 error of this tensor against fp64 (not stored for the full-size fixture: its extractor tensors
    # get the config-1 level, 4e-3)
    e_own = float(own[i]) if own is not None else (4e-3 if name.startswith('feature_extraction') else 2e-4)
    bound = max(1e-3, 5.0 * e_own)
    worst = max(worst, rel)
    if not name.startswith('feature_extraction'):
      worst_3d = max(worst_3d, rel)
    assert rel <= 1.5 * bound, (name, rel, bound)  # (16 projections: the estimate itself scatters by ~ +-35 %)
    # 32 sampled entries: round-off of a back-propagated gradient is heavy-tailed over the entries of a tensor (the reference's own
    # fp32 run against fp64: median 7e-5 rms, 99th percentile 1e-3 rms, maximum 1e-2 rms at config 1), hence a robust pair of
    # bounds -- the median at the L2 level, every entry within 100x of it (0.1 rms at most: a wrong entry is O(1) rms)
    idx = z['train/grad_idx'][i]
    rms = norm / np.sqrt(g.size)
    diff = np.abs(g[idx] - z['train/grad_val'][i])
    assert np.median(diff) <= 3.0 * bound * rms + 1e-12, (name, 'sampled entries: median', float(np.median(diff)), bound * rms)
    assert diff.max() <= 100.0 * bound * rms + 1e-12, (name, 'sampled entries: max', float(diff.max()), bound * rms)
  print('%s: relative L2 error of the parameter gradients (243 tensors): worst %.3e, worst outside the extractor %.3e' % (tag, worst, worst_3d))


@pytest.mark.parametrize('tag', ['tiny', 'cfg1', 'full'])
def test_eval_output_within_1e3_of_the_reference(golden, tag):
  z = golden('model_wc_%s.npz' % tag)
  net, left, right, gt, seed = _load(z, bn_from_fixture=True)
  net.eval()
  net.out_conf = True
  with torch.no_grad():
    pred, conf = net(left, right)
  _check_pred('%s eval pred3' % tag, pred, z, 'eval/pred3', z['truth64/eval_E_ref'])
  sub = int(z['sub'])
  # confidence = P(round(d) - 1) + P(round(d)) + P(round(d) + 1): compare where round(d) is not at a tie
  ref_pred = z['eval/pred3']
  stable = np.abs(np.abs(ref_pred - np.round(ref_pred)) - 0.5) > 0.01
  diff = np.abs(conf[:, :, ::sub, ::sub].cpu().numpy() - z['eval/conf'])
  assert diff[stable].max() < 1e-3, diff[stable].max()
