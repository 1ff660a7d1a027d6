#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *imported reference* on the CPU.

Runs ONLY in the development container (needs /root/reference); nothing here is used at test
time -- the tests read the committed .npz / .json files.  The reference Python is imported
unmodified with three harness-side shims:

  1. ``models.basic.spherical_conv.sphere_conv_cuda`` (the un-shipped compiled extension) is
     pre-seeded with an empty module so that sphere_conv.py:12 resolves;
  2. ``torch.Tensor.cuda`` / ``nn.Module.cuda`` become the identity (hard-coded .cuda() calls);
  3. the module-global ``sphere_conv`` (looked up at call time by SphereConv.forward,
     sphere_conv.py:243) is rebound to ``oracle.sphere_conv_ref.sphere_conv``, because
     SphereConvFunction.forward rejects CPU tensors and no CPU source of the native op exists.

Usage:  python tests/golden/make_golden.py [--skip-cfg1]
"""
import argparse
import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('MODE_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import sphere_conv_ref  # noqa: E402
import recipe  # noqa: E402

warnings.filterwarnings('ignore')


def import_reference():
  sys.modules['models.basic.spherical_conv.sphere_conv_cuda'] = types.ModuleType('sphere_conv_cuda')
  torch.Tensor.cuda = lambda self, *a, **k: self
  nn.Module.cuda = lambda self, *a, **k: self
  sys.path.insert(0, REF)
  import models  # the reference's package
  import models.basic.spherical_conv.sphere_conv as sc
  sc.sphere_conv = sphere_conv_ref.sphere_conv
  assert os.path.realpath(models.__file__).startswith(os.path.realpath(REF))
  return models, sc


def sha(a):
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_recipe(model, seed):
  manifest = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
  model.load_state_dict(recipe.recipe_state(manifest, seed))
  return manifest


def grad_summary(model, n_samples=4, seed=0):
  rs = np.random.RandomState(seed)
  names, sums, idx, vals = [], [], [], []
  for k, p in model.named_parameters():
    g = p.grad.detach().reshape(-1).double()
    names.append(k)
    sums.append(float(g.abs().sum()))
    ii = rs.randint(0, g.numel(), n_samples)
    idx.append(ii)
    vals.append(g[ii].numpy())
  return dict(grad_names=np.array(names), grad_abs_sum=np.array(sums), grad_idx=np.array(idx),
              grad_val=np.array(vals))


def bn_stats(model):
  out = {}
  for k, v in model.state_dict().items():
    if k.endswith('running_mean') or k.endswith('running_var'):
      out['bn/' + k] = v.numpy().copy()
  return out


def gen_positions(sc):
  out = {}
  meta = {}
  for typ, (ih, iw) in [('ERP', (8, 16)), ('Cassini', (16, 8))]:
    m = sc.SphereConv(ih, iw, typ, 1, 1, 3, 1, 1, 1, 1, False)
    out['%s_%dx%d' % (typ, ih, iw)] = m.position.numpy()
  for typ, (ih, iw) in [('Cassini', (256, 128)), ('Cassini', (128, 64)), ('ERP', (128, 256)), ('Cassini', (16, 8)),
                        ('Cassini', (512, 256))]:
    m = sc.SphereConv(ih, iw, typ, 1, 1, 3, 1, 1, 1, 1, False)
    p = m.position.numpy()
    rs = np.random.RandomState(1)
    ii = rs.randint(0, p.size, 1024)
    key = '%s_%dx%d' % (typ, ih, iw)
    meta[key] = dict(shape=list(p.shape), sha256=sha(p), min=float(p.min()), max=float(p.max()))
    out[key + '_idx'] = ii
    out[key + '_val'] = p.reshape(-1)[ii]
  np.savez_compressed(os.path.join(HERE, 'positions.npz'), **out)
  with open(os.path.join(HERE, 'positions_meta.json'), 'w') as f:
    json.dump(meta, f, indent=1)
  print('positions', {k: v['sha256'][:12] for k, v in meta.items()})


def gen_sphere_conv(sc):
  """G2.  The native op has no CPU source: these vectors come from the restatement (direct form,
  evaluated in fp64 on fp32-representable inputs) and are cross-checked here against the
  grid_sample form.  They pin the oracle against drift, not against the reference."""
  out = {}
  cases = [('erp_s1', 'ERP', 16, 32, 3, 4, 1, 1), ('cas_s1', 'Cassini', 32, 16, 8, 8, 1, 1),
           ('erp_s2', 'ERP', 16, 32, 4, 6, 2, 1), ('cas_g2', 'Cassini', 32, 16, 4, 4, 1, 2)]
  for name, typ, ih, iw, ci, co, s, g in cases:
    m = sc.SphereConv(ih, iw, typ, ci, co, 3, s, 1, 1, g, False)
    pos = m.position
    H, W = pos.shape[2:]
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal((2, ci, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((co, ci // g, 3, 3)) * 0.2).astype(np.float32))
    y = sphere_conv_ref.forward(x.double(), pos, w.double(), (s, s), (1, 1), (1, 1), g)
    y2 = sphere_conv_ref.forward_grid_sample(x.double(), pos, w.double(), (s, s), (1, 1), (1, 1), g)
    assert (y - y2).abs().max() < 1e-12, (name, float((y - y2).abs().max()))
    gy = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
    gx, gw = sphere_conv_ref.backward(x.double(), pos, w.double(), gy.double(), (s, s), (1, 1), (1, 1), g)
    xa = x.double().requires_grad_(True)
    wa = w.double().requires_grad_(True)
    sphere_conv_ref.forward_grid_sample(xa, pos, wa, (s, s), (1, 1), (1, 1), g).backward(gy.double())
    assert (gx - xa.grad).abs().max() < 1e-11 and (gw - wa.grad).abs().max() < 1e-10
    # what the reference module's own forward produces through shim 3 (fp32)
    with torch.no_grad():
      m.weight.copy_(w)
      y32 = m(x)
    assert (y32.double() - y).abs().max() < 1e-4
    for k, v in dict(x=x, w=w, gy=gy, y=y, gx=gx, gw=gw).items():
      out['%s/%s' % (name, k)] = v.numpy()
    out['%s/cfg' % name] = np.array([ih, iw, ci, co, s, g])
    out['%s/type' % name] = np.array(typ)
  np.savez_compressed(os.path.join(HERE, 'sphere_conv.npz'), **out)
  print('sphere_conv cases', [c[0] for c in cases])


def gen_hourglass(models):
  from models.mode_disparity import hourglass
  torch.manual_seed(0)
  hg = hourglass(4)
  manifest = load_recipe(hg, 11)
  hg.train()
  rs = np.random.RandomState(3)
  out = dict(manifest=np.array(json.dumps([[k, list(s)] for k, s in manifest])))
  x = torch.from_numpy(rs.standard_normal((2, 4, 8, 8, 8)).astype(np.float32))
  pre_in = torch.from_numpy(rs.standard_normal((2, 8, 4, 4, 4)).astype(np.float32))
  post_in = torch.from_numpy(rs.standard_normal((2, 8, 4, 4, 4)).astype(np.float32))
  out.update(x=x.numpy(), presqu=pre_in.numpy(), postsqu=post_in.numpy())
  for tag, (a, b) in dict(none=(None, None), both=(pre_in, post_in)).items():
    hg.zero_grad()
    xa = x.clone().requires_grad_(True)
    o, pre, post = hg(xa, a, b)
    go = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(o.shape)).astype(np.float32))
    (o * go).sum().backward()
    out.update({tag + '/out': o.detach().numpy(), tag + '/pre': pre.detach().numpy(), tag + '/post': post.detach().numpy(),
                tag + '/gout': go.numpy(), tag + '/gx': xa.grad.numpy()})
    for k, p in hg.named_parameters():
      out['%s/grad/%s' % (tag, k)] = p.grad.numpy().copy()
  np.savez_compressed(os.path.join(HERE, 'hourglass.npz'), **out)
  print('hourglass ok')


def run_model(models, maxdisp, H, W, B, seed, tag, full_outputs, with_grad=True):
  torch.manual_seed(0)
  m = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini', out_conf=False)
  manifest = load_recipe(m, seed)
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity(B, H, W, seed + 2, maxdisp)
  mask = ~torch.isnan(gt)
  out = dict(cfg=np.array([maxdisp, H, W, B, seed]))
  taps = {}
  hooks = [
      m.feature_extraction.register_forward_hook(lambda mod, i, o: taps.setdefault('fea', []).append(o.detach())),
      m.dres0.register_forward_pre_hook(lambda mod, i: taps.__setitem__('cost', i[0].detach())),
      m.classif3.register_forward_hook(lambda mod, i, o: taps.__setitem__('classif3', o.detach())),
  ]
  # ---- train-mode forward (+ backward): batch statistics
  m.train()
  p1, p2, p3 = m(left, right)
  import torch.nn.functional as F
  loss = 0.5 * F.smooth_l1_loss(p1[mask], gt[mask]) + 0.7 * F.smooth_l1_loss(p2[mask], gt[mask]) + \
      F.smooth_l1_loss(p3[mask], gt[mask])
  out['train/loss'] = np.array(float(loss))
  sub = (slice(None),) * 2 + ((slice(None), slice(None)) if full_outputs else (slice(None, None, 4), slice(None, None, 4)))
  for i, p in enumerate((p1, p2, p3)):
    out['train/pred%d' % (i + 1)] = p.detach()[sub].numpy()
    out['train/pred%d_mean' % (i + 1)] = np.array(float(p.detach().double().mean()))
  out['train/fea_left'] = taps['fea'][0][:, :, ::(1 if full_outputs else 4), ::(1 if full_outputs else 4)].numpy()
  c = taps['cost']
  out['train/cost_sha256'] = np.array(sha(c.numpy()))
  out['train/cost_sum'] = np.array(float(c.double().sum()))
  if full_outputs:
    out['train/cost'] = c.numpy()
    out['train/fea_right'] = taps['fea'][1].numpy()
  if with_grad:
    loss.backward()
    out.update({'train/' + k: v for k, v in grad_summary(m).items()})
  # ---- calibrate BN running statistics with one more train-mode forward, then eval
  # (momentum 1.0 for this pass => running stats := this batch's statistics, so that eval mode is as
  # well conditioned as train mode; SURVEY.md section 0 fact 5.  Module attribute only, no reference edit.)
  taps.clear()
  bns = [x for x in m.modules() if isinstance(x, (nn.BatchNorm2d, nn.BatchNorm3d))]
  for x in bns:
    x.momentum = 1.0
  with torch.no_grad():
    m(left, right)
  for x in bns:
    x.momentum = 0.1
  out.update(bn_stats(m))
  m.eval()
  m.out_conf = True
  taps.clear()
  with torch.no_grad():
    pred, conf = m(left, right)
  out['eval/pred3'] = pred[sub].numpy()
  out['eval/conf'] = conf.unsqueeze(1)[sub].numpy() if conf.dim() == 3 else conf[sub].numpy()
  out['eval/pred3_mean'] = np.array(float(pred.double().mean()))
  out['eval/logits3'] = taps['classif3'].numpy() if full_outputs else taps['classif3'][:, :, ::2, ::4, ::4].numpy()
  for h in hooks:
    h.remove()
  # ---- fp64 evaluation of the same network by the (pinned) oracle: the reference's own fp32 run is only reproducible
  # to E_ref = max|ref32 - truth64| (1e-3 .. 1e-2 here), which is what an independent fp32 implementation can be held to.
  from oracle import mode_ref
  P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in recipe.recipe_state(manifest, seed).items()}
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
  with torch.no_grad():
    t64 = mode_ref.mode_disparity(P64, left.double(), right.double(), maxdisp, pos, True)
  for i, p in enumerate(t64):
    out['truth64/train_pred%d' % (i + 1)] = p[sub].numpy()
  out['truth64/train_E_ref'] = np.array(max(float((torch.from_numpy(out['train/pred%d' % (i + 1)]).double() - t64[i][sub]).abs().max())
                                            for i in range(3)))
  for k, v in out.items():
    if k.startswith('bn/'):
      P64[k[3:]] = torch.from_numpy(v).double()
  with torch.no_grad():
    e64 = mode_ref.mode_disparity(P64, left.double(), right.double(), maxdisp, pos, False)
  out['truth64/eval_pred3'] = e64[sub].numpy()
  out['truth64/eval_E_ref'] = np.array(float((torch.from_numpy(out['eval/pred3']).double() - e64[sub]).abs().max()))
  print('  E_ref (reference fp32 vs fp64): train %.3e eval %.3e' % (float(out['truth64/train_E_ref']), float(out['truth64/eval_E_ref'])))
  np.savez_compressed(os.path.join(HERE, 'model_%s.npz' % tag), **out)
  print('model', tag, 'loss', float(loss), 'eval mean', float(pred.mean()),
        'frac integer', float((pred == pred.round()).float().mean()))
  return manifest


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--skip-cfg1', action='store_true')
  ap.add_argument('--only', default='')
  args = ap.parse_args()
  torch.set_num_threads(8)
  models, sc = import_reference()
  todo = args.only.split(',') if args.only else ['positions', 'sphere_conv', 'hourglass', 'tiny', 'cfg1']
  if 'positions' in todo:
    gen_positions(sc)
  if 'sphere_conv' in todo:
    gen_sphere_conv(sc)
  if 'hourglass' in todo:
    gen_hourglass(models)
  if 'tiny' in todo:
    manifest = run_model(models, 16, 64, 32, 2, 100, 'tiny', full_outputs=True)
    with open(os.path.join(HERE, 'manifest_mode_disparity.json'), 'w') as f:
      json.dump([[k, list(s)] for k, s in manifest], f)
  if 'cfg1' in todo and not args.skip_cfg1:
    run_model(models, 64, 512, 256, 1, 200, 'cfg1', full_outputs=False, with_grad=True)


if __name__ == '__main__':
  main()
