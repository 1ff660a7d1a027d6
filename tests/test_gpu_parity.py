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
  * parameter gradients: relative L2 error per tensor <= max(floor, 5 x the reference's own fp32-vs-fp64 error of that tensor),
    floor = 1e-3 for the 3-D stage, 4e-3 / 1e-2 for the extractor's convolution weights / BatchNorm vectors (_grad_floor), and
    the L2 error over ALL parameters <= max(1e-3, 3 x own), over the whole extractor <= max(2e-3, 3 x own),
    estimated from 16 Rademacher projections stored in the fixture (E <e, v>^2 = |e|^2), plus 32 sampled entries per tensor
    (median at the L2 level, maximum within 100x of it).
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
GRAD_CAP = 5e-2  # absolute cap on the relative L2 error of any parameter-gradient tensor


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
  assert torch.cuda.is_available(), 'GPU tests need a GPU'
  mode_hip.lib()


def _load(z, bn_from_fixture=False):
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  sd = recipe.fixture_state(z)
  if bn_from_fixture:
    for k in z.files:
      if k.startswith('bn/'):
        sd[k[3:]] = torch.from_numpy(z[k]).clone()
  net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini').to(DEV)
  net.load_state_dict(sd)
  left, right, gt = recipe.fixture_inputs(z)
  return net, left.to(DEV), right.to(DEV), gt.to(DEV), seed


def _grad_floor(name, ndim, tiny=False, peaked_full=False):
  """Relative L2 level a parameter gradient is held to at least.
    3-D stage (dres*, classif*): 1e-3.
    extractor: its gradients are what is left of O(1) terms after ~60 BatchNorm backward passes cancel all but ~1e-3 of them
    (grad_norm 1e-3 .. 5e-2 for its BatchNorm vectors, ~5 for its convolution weights, ~1 .. 40 in the 3-D stage).  Measured in
    fp64 on the 64 x 32 fixture: perturbing the weights by 1e-7 relative -- one fp32 rounding each -- moves the extractor's
    BatchNorm gradients by up to 1.1e-3 and its convolution-weight gradients by up to 4e-4, the 3-D stage's by 2.5e-6; the
    reference's own fp32 run against fp64 shows 1e-3 .. 4e-3 on extractor tensors at config 1 and up to 1.7e-2 at full size.
    Hence 4e-3 for the extractor's convolution weights and 1e-2 for its BatchNorm vectors; a dropped or wrong term is O(1), and
    the whole-network and whole-extractor L2 errors are held to max(1e-3 / 2e-3, 3 x the reference's own) on top (_check_grads)."""
  if not name.startswith('feature_extraction'):
    # The peaked-softmax fixture at the benchmark size: the loss gradient is concentrated on few voxels, and the ReLU masks of two
    # correct fp32 evaluations differ in 1 .. 65 of the 1.5 M .. 50 M elements of every layer of the 3-D stage (counted between this
    # implementation's own two arithmetics with tools/experiments/arith_divergence.py: e.g. 16 of 12.6 M at dres3.conv2) -- each such
    # element moves a per-channel sum by a whole term.  The reference's own fp32 run shows the same against float64 (up to 3.6e-3 on
    # dres2.conv3 / dres2.conv1 tensors, 1e-4 on others: which tensors are hit is chance).  Hence 4e-3 for the per-channel BatchNorm
    # vectors and 2e-3 for the weights there; everywhere else the 3-D stage is held to 1e-3.
    if peaked_full:
      return 2e-3 if ndim > 1 else 4e-3
    return 1e-3
  # At the 64 x 32 fixture a layer of the extractor has 512 pixels per channel, and ONE ReLU whose pre-activation lies within a
  # rounding of zero decides ~1/500 of that layer's gradients: layer3.2.conv1 has such an element (-1.3e-6 with the fp32 MFMA
  # kernels, +9.6e-7 on the split-bf16 path, the two pre-activation tensors agreeing to 2.8e-5 everywhere: tools/debug_relu_flip.py),
  # which moves that layer's weight gradient by 6.2e-3 and its BatchNorm bias gradient by 1.2e-2 while every other tensor stays
  # below 1e-3.  The reference's own fp32 run happens to land on the fp64 side of that zero.  Twice the floors at that size.
  k = 2.0 if tiny else 1.0
  return k * (4e-3 if ndim > 1 else 1e-2)


def _check_grads(tag, net, z, seed):
  grads = dict(net.named_parameters())
  own = z['truth64/grad_rel_l2'] if 'truth64/grad_rel_l2' in z.files else None
  K = z['train/grad_proj'].shape[1]
  worst, worst_3d = 0.0, 0.0
  d_all, d_ext = np.zeros(K), np.zeros(K)
  n_all = n_ext = o_all = o_ext = 0.0  # squared norms of the reference gradient and of the reference's own fp32 error
  for i, name in enumerate(z['train/grad_names']):
    name = str(name)
    p = grads[name]
    g = p.grad.detach().cpu().reshape(-1).double().numpy()
    norm = float(z['train/grad_norm'][i])
    proj = recipe.projection_signs(seed, i, g.size, K).astype(np.float64) @ g
    delta = proj - z['train/grad_proj'][i]
    rel = float(np.sqrt(np.mean(delta**2))) / (norm + 1e-300)
    e_own = float(own[i]) if own is not None else 2e-4
    bound = max(_grad_floor(name, p.dim(), tag.startswith('tiny'), tag.startswith('peaked full')), 5.0 * e_own)
    worst = max(worst, rel)
    d_all += delta
    n_all += norm**2
    o_all += (e_own * norm)**2
    if name.startswith('feature_extraction'):
      d_ext += delta
      n_ext += norm**2
      o_ext += (e_own * norm)**2
    else:
      worst_3d = max(worst_3d, rel)
    # (16 projections: the estimate itself scatters by ~ +-35 %); whatever the reference's own error, never more than 5 % of a tensor
    assert rel <= min(1.5 * bound, GRAD_CAP), (name, rel, bound)
    # 32 sampled entries: round-off of a back-propagated gradient is heavy-tailed over the entries of a tensor (the reference's own
    # fp32 run against fp64: median 7e-5 rms, 99th percentile 1e-3 rms, maximum 1e-2 rms at config 1), hence a robust pair of
    # bounds -- the median at the L2 level, every entry within 100x of it (a wrong entry is O(1) rms)
    idx = z['train/grad_idx'][i]
    rms = norm / np.sqrt(g.size)
    diff = np.abs(g[idx] - z['train/grad_val'][i])
    assert np.median(diff) <= 3.0 * bound * rms + 1e-12, (name, 'sampled entries: median', float(np.median(diff)), bound * rms)
    assert diff.max() <= 100.0 * bound * rms + 1e-12, (name, 'sampled entries: max', float(diff.max()), bound * rms)
  rel_all = float(np.sqrt(np.mean(d_all**2)) / np.sqrt(n_all))
  rel_ext = float(np.sqrt(np.mean(d_ext**2)) / np.sqrt(n_ext))
  print('%s: relative L2 error of the parameter gradients: whole network %.2e, whole extractor %.2e; per tensor (243): worst %.2e, '
        'worst outside the extractor %.2e' % (tag, rel_all, rel_ext, worst, worst_3d))
  # flattened over all parameters / over the extractor: at most 3 x the reference's own fp32 error of the same vector
  own_all, own_ext = float(np.sqrt(o_all / n_all)), float(np.sqrt(o_ext / n_ext))
  assert rel_all <= max(1e-3, 3.0 * own_all), (rel_all, own_all)
  assert rel_ext <= max(2e-3, 3.0 * own_ext), (rel_ext, own_ext)


def _check_pred(name, got, z, key, e_ref, tol=DISP_TOL):
  sub = int(z['sub'])
  g = got.detach()
  d_pix = np.abs(g[:, :, ::sub, ::sub].cpu().numpy().astype(np.float64) - z[key])
  d_blk = np.abs(F.avg_pool2d(g.double(), 8).cpu().numpy() - z[key + '_block'])
  print('%s: max|HIP - reference fp32| = %.3e px (stored pixels), %.3e (8x8 block means, all pixels); reference vs fp64: %.3e' %
        (name, d_pix.max(), d_blk.max(), float(e_ref)))
  assert d_pix.max() <= tol, (name, d_pix.max())
  assert d_blk.max() <= tol, (name, d_blk.max())
  if tol > DISP_TOL:  # a bound widened by the reference's own error at a few multi-modal pixels: 99.9 % of the pixels within 1e-3 all the same
    q = float(np.quantile(d_pix, 0.999))
    print('%s: 99.9th percentile of |HIP - reference fp32| = %.3e px' % (name, q))
    t64 = 'truth64/' + key.replace('/', '_')
    if q > DISP_TOL and t64 in z.files:
      # Two fp32 evaluations that are each within E of the exact network differ by up to 2 E: where the reference's OWN 99.9th
      # percentile against float64 is close to 1e-3 (7.6e-4 / 8.1e-4 px at the benchmark size with a peaked softmax: a handful of
      # multi-modal border pixels), the comparison has to be made against float64 -- the HIP path must be as close to the exact
      # network as the reference is (within 1.5 x at this percentile; 8 192 stored pixels: the statistic is the 8th largest value, itself noisy).
      truth = z[t64]
      own = float(np.quantile(np.abs(z[key].astype(np.float64) - truth), 0.999))
      mine = float(np.quantile(np.abs(g[:, :, ::sub, ::sub].cpu().numpy().astype(np.float64) - truth), 0.999))
      print('%s: 99.9th percentile against float64: HIP %.3e px, the reference itself %.3e px' % (name, mine, own))
      assert mine <= max(DISP_TOL, 1.5 * own), (name, mine, own)
    else:
      assert q <= DISP_TOL, (name, q)


@pytest.mark.parametrize('tag', ['tiny', 'cfg1', 'full'])
def test_train_outputs_and_gradients_within_1e3_of_the_reference(golden, tag, arith):
  z = golden('model_wc_%s.npz' % tag)
  net, left, right, gt, seed = _load(z)
  net.train()
  preds = net(left, right)
  for i, p in enumerate(preds):
    _check_pred('%s [%s] train pred%d' % (tag, arith, i + 1), p, z, 'train/pred%d' % (i + 1), z['truth64/train_E_ref'])
  loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
  ref_loss = float(z['train/loss'])
  assert abs(float(loss.detach()) - ref_loss) <= 2e-5 * ref_loss, (float(loss.detach()), ref_loss)
  loss.backward()
  _check_grads('%s [%s]' % (tag, arith), net, z, seed)


@pytest.mark.parametrize('tag', ['tiny', 'cfg1', 'full'])
def test_eval_output_within_1e3_of_the_reference(golden, tag, arith):
  z = golden('model_wc_%s.npz' % tag)
  net, left, right, gt, seed = _load(z, bn_from_fixture=True)
  net.eval()
  net.out_conf = True
  with torch.no_grad():
    pred, conf = net(left, right)
  _check_pred('%s [%s] eval pred3' % (tag, arith), pred, z, 'eval/pred3', z['truth64/eval_E_ref'])
  sub = int(z['sub'])
  # confidence = P(round(d) - 1) + P(round(d)) + P(round(d) + 1): compare where round(d) is not at a tie
  ref_pred = z['eval/pred3']
  stable = np.abs(np.abs(ref_pred - np.round(ref_pred)) - 0.5) > 0.01
  diff = np.abs(conf[:, :, ::sub, ::sub].cpu().numpy() - z['eval/conf'])
  assert diff[stable].max() < 1e-3, diff[stable].max()


def test_eval_output_at_config4_size_against_the_float64_oracle(golden):
  """BASELINE configs[4] with an ORACLE ("per-GPU share exercised by property checks only"): the inference forward of one
  2048 x 1024 pair at 256 disparities -- test_disparity.py:120-154 at the largest size the north_star names -- against
  tests/golden/model_wc_config4_eval.npz, the float64 evaluation of oracle/mode_ref.py stored by make_golden_config4_eval.py (nine CPU
  minutes on the box's host cores: stored, not recomputed per run; recomputed live once in round 6: max 2.8e-4 px).  Well-conditioned
  recipe weights, running statistics from the fixture (the batch statistics of the pair, float32).  On the default arithmetic of an eval
  forward (two fp16 pieces in the stride-1 3-D and 3 x 3 layers) AND on three bf16 pieces, under the no-vendor guard; every
  stored pixel and every 8 x 8 block mean within the north_star's 1e-3 px.  And the three predictions of the TRAINING-mode forward of the
  same pair (the fixture's calibration pass) on a training step's arithmetic: the forward half of configs[4]'s step, same bound (its
  backward stays on the size-independent properties of test_gpu_fullsize.py: a float64 backward of the oracle at this size needs > 300 GB)."""
  from mode_hip import no_vendor
  from mode_hip import functional as HF
  z = golden('model_wc_config4_eval.npz')
  net, left, right, gt, seed = _load(z, bn_from_fixture=True)
  net.eval()
  keep = HF.CONV3D_EVAL_F16, HF.CONV2D_EVAL_F16
  try:
    for f16 in (True, False):
      HF.CONV3D_EVAL_F16 = HF.CONV2D_EVAL_F16 = f16
      with torch.no_grad(), no_vendor.no_vendor_arithmetic():
        pred = net(left, right)
      _check_pred('config4 eval pred3 [%s]' % ('two fp16 pieces' if f16 else 'three bf16 pieces'), pred, z, 'eval/pred3', 0.0)
  finally:
    HF.CONV3D_EVAL_F16, HF.CONV2D_EVAL_F16 = keep
  # the training-mode forward of the same pair (batch statistics; the generator's calibration pass): its three predictions, on the
  # arithmetic of a training step (two fp16 pieces in the stride-1 3-D, spherical and 3 x 3 layers)
  net.train()
  with torch.no_grad(), no_vendor.no_vendor_arithmetic():
    preds = net(left, right)
  for i, p in enumerate(preds):
    _check_pred('config4 train-mode forward pred%d' % (i + 1), p, z, 'train/pred%d' % (i + 1), 0.0)


def test_config2_batch_of_two_at_full_size(golden, arith):
  """BASELINE configs[2]: 1024 x 512, 192 disparities, batch 2, forward + backward.  The reference fixture holds ONE pair (a CPU
  run of two costs minutes and ~50 GB); a batch of two copies of that pair has the same BatchNorm statistics as the pair alone, so
  every sample of the batch-2 step must reproduce the reference's batch-1 outputs to the north_star's 1e-3, and the parameter
  gradients (mean over twice the pixels of twice the terms) the batch-1 gradients."""
  z = golden('model_wc_full.npz')
  net, left, right, gt, seed = _load(z)
  net.train()
  left2, right2, gt2 = [torch.cat((t, t), 0) for t in (left, right, gt)]
  preds = net(left2, right2)
  for i, p in enumerate(preds):
    assert tuple(p.shape) == (2, 1, 1024, 512)
    for b in range(2):
      _check_pred('configs[2] sample %d pred%d' % (b, i + 1), p[b:b + 1], z, 'train/pred%d' % (i + 1), z['truth64/train_E_ref'])
  loss = mode_ref.training_loss(preds, gt2, ~torch.isnan(gt2))
  assert abs(float(loss.detach()) - float(z['train/loss'])) <= 2e-5 * float(z['train/loss'])
  loss.backward()
  _check_grads('configs[2] batch-2 step', net, z, seed)


# ------------------------------------------------------------------ the same bar where it is hard: a peaked (trained) softmax
# ('full' = the benchmark size 1024 x 512 / 192, one pair: VERDICT r3 item 7 -- the 1e-3 bar in the peaked regime at the size that is timed)
@pytest.mark.parametrize('tag', ['tiny', 'cfg1', 'full'])
def test_peaked_softmax_train_outputs_and_gradients(golden, tag, arith):
  """tests/golden/model_peaked_*.npz: the classifier heads were TRAINED by the imported reference until the softmax over the
  disparity axis holds most of its mass within +-1 px of the prediction (mean confidence 0.63 / 0.96 against 3/D = 0.19 / 0.05 for
  the near-uniform softmax of the model_wc_* fixtures) -- the regime of a trained network, where d(disparity)/d(logit) is not
  damped by a flat distribution.  Bound: max(1e-3, 3 x E_ref) with E_ref = the reference's own fp32-vs-fp64 error on the fixture,
  printed so that the regime is visible."""
  z = golden('model_peaked_%s.npz' % tag)
  conf_mean, e_ref = float(z['eval/conf'].mean()), float(z['truth64/train_E_ref'])
  assert conf_mean >= 0.5
  tol = max(DISP_TOL, 3.0 * e_ref)
  print('peaked %s: mean confidence of the reference output %.3f (a uniform softmax has %.3f); E_ref(train) = %.3e px -> bound %.3e px' %
        (tag, conf_mean, 3.0 / int(z['cfg'][0]), e_ref, tol))
  net, left, right, gt, seed = _load(z)
  net.train()
  preds = net(left, right)
  for i, p in enumerate(preds):
    _check_pred('peaked %s [%s] train pred%d' % (tag, arith, i + 1), p, z, 'train/pred%d' % (i + 1), e_ref, tol)
  loss = mode_ref.training_loss(preds, gt, ~torch.isnan(gt))
  ref_loss = float(z['train/loss'])
  assert abs(float(loss.detach()) - ref_loss) <= 2e-5 * ref_loss, (float(loss.detach()), ref_loss)
  loss.backward()
  _check_grads('peaked %s [%s]' % (tag, arith), net, z, seed)


# ('full' = the benchmark size 1024 x 512 / 192, one pair: VERDICT r3 item 7 -- the 1e-3 bar in the peaked regime at the size that is timed)
@pytest.mark.parametrize('tag', ['tiny', 'cfg1', 'full'])
def test_peaked_softmax_eval_output_and_confidence(golden, tag, arith):
  z = golden('model_peaked_%s.npz' % tag)
  e_ref = float(z['truth64/eval_E_ref'])
  tol = max(DISP_TOL, 3.0 * e_ref)
  net, left, right, gt, seed = _load(z, bn_from_fixture=True)
  net.eval()
  net.out_conf = True
  with torch.no_grad():
    pred, conf = net(left, right)
  print('peaked %s: E_ref(eval) = %.3e px -> bound %.3e px' % (tag, e_ref, tol))
  _check_pred('peaked %s [%s] eval pred3' % (tag, arith), pred, z, 'eval/pred3', e_ref, tol)
  sub = int(z['sub'])
  ref_pred = z['eval/pred3']
  stable = np.abs(np.abs(ref_pred - np.round(ref_pred)) - 0.5) > 0.01
  diff = np.abs(conf[:, :, ::sub, ::sub].cpu().numpy() - z['eval/conf'])
  assert diff[stable].max() < 1e-3, diff[stable].max()
  assert float(conf.mean()) >= 0.5
