#!/usr/bin/env python3
"""K-step training trajectory of the *imported reference*: tests/golden/model_steps_tiny.npz.

Every other fixture pins ONE forward / backward.  The reference's training loop (train_disparity.py:147-161, trainDisp) is
zero_grad -> forward -> masked smooth-L1 loss 0.5 / 0.7 / 1.0 -> backward -> optimizer.step() repeated, with Adam(lr 1e-3, betas (0.9, 0.999))
(train_disparity.py:287); the product replays zero-grad + forward + loss + backward as a hipGraph, lets its kernels add into a flat gradient
buffer and steps a fused Adam.  This script runs K = 3 iterations of the reference's own loop body on the CPU (same harness as
make_golden.py: development container only, three shims, no reference file edited) at 64 x 32 / 16 disparities, batch 2, from the
well-conditioned recipe state, and stores what a replacement must reproduce:

  cfg                  [maxdisp, H, W, B, seed, K]
  loss                 the K loss values (float64 of the reference's float32 loss.item())
  bn/<key>             every BatchNorm running_mean / running_var after step K;  nbt = num_batches_tracked after step K: [3-D stage, extractor]
  names, idx           parameter names in named_parameters() order and 32 sampled flat indices per parameter
  p0_val, pK_val       the sampled entries before step 1 and after step K
  dnorm                ||p_K - p_0||_2 per parameter;  dproj = 16 Rademacher projections of (p_K - p_0) per parameter (recipe.projection_signs)
  loss64, bn64/<key>, dproj64   the same trajectory by the float64 oracle: |reference fp32 - fp64| is the reference's own sensitivity (E_ref)

Usage:  python tests/golden/make_golden_steps.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import recipe  # noqa: E402

K_STEPS, K_PROJ, N_SAMPLES = 3, 16, 32


def main():
  models, _ = mg.import_reference()
  maxdisp, H, W, B, seed = 16, 64, 32, 2, 300
  torch.manual_seed(0)
  torch.set_num_threads(8)
  net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini', out_conf=False)
  manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
  assert manifest == recipe.load_manifest()
  net.load_state_dict(recipe.recipe_state_wc(manifest, seed))
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity_smooth(B, H, W, seed + 2, maxdisp)
  mask = ~torch.isnan(gt)
  opt = torch.optim.Adam(net.parameters(), lr=0.001, betas=(0.9, 0.999))  # train_disparity.py:287
  names = [k for k, _ in net.named_parameters()]
  p0 = [p.detach().clone() for p in net.parameters()]
  losses = []
  for _ in range(K_STEPS):  # the body of trainDisp, train_disparity.py:147-161 (size_average=True is reduction='mean')
    net.train()
    opt.zero_grad()
    o1, o2, o3 = net(left, right)
    loss = 0.5 * F.smooth_l1_loss(o1[mask], gt[mask]) + 0.7 * F.smooth_l1_loss(o2[mask], gt[mask]) + F.smooth_l1_loss(o3[mask], gt[mask])
    loss.backward()
    opt.step()
    losses.append(float(loss.data.item()))
    print('step %d: loss %.8f' % (len(losses), losses[-1]))
  out = dict(cfg=np.array([maxdisp, H, W, B, seed, K_STEPS]), loss=np.array(losses, dtype=np.float64), names=np.array(names))
  idx, v0, vk, dnorm, dproj = [], [], [], [], []
  for i, (p, q) in enumerate(zip(net.parameters(), p0)):
    a, b = p.detach().reshape(-1).double().numpy(), q.reshape(-1).double().numpy()
    ii = np.random.RandomState(seed + 7919 * (i + 1)).randint(0, a.size, N_SAMPLES)
    idx.append(ii)
    v0.append(b[ii])
    vk.append(a[ii])
    dnorm.append(float(np.sqrt(((a - b)**2).sum())))
    dproj.append(recipe.projection_signs(seed, i, a.size, K_PROJ).astype(np.float64) @ (a - b))
  out.update(idx=np.array(idx), p0_val=np.array(v0), pK_val=np.array(vk), dnorm=np.array(dnorm), dproj=np.array(dproj))
  # The reference's OWN sensitivity: the same three iterations by the float64 oracle (oracle/mode_ref.py in double, same optimizer).  Adam's
  # first steps move every entry by ~lr * sign(g), whatever |g|: where a gradient is within fp32 round-off of zero the direction is decided
  # by that round-off, so two exact fp32 evaluations follow visibly different trajectories.  |fp32 - fp64| of the reference is the yardstick
  # the GPU test holds the product to (E_ref, as in the other fixtures).
  from oracle import mode_ref
  P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in recipe.recipe_state_wc(manifest, seed).items()}
  for k in names:
    P[k].requires_grad_(True)
  p0d = {k: P[k].detach().clone() for k in names}
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini').double()
  opt64 = torch.optim.Adam([P[k] for k in names], lr=0.001, betas=(0.9, 0.999))
  loss64 = []
  for _ in range(K_STEPS):
    opt64.zero_grad()
    l64 = mode_ref.training_loss(mode_ref.mode_disparity(P, left.double(), right.double(), maxdisp, pos, True), gt.double(), mask)
    l64.backward()
    opt64.step()
    loss64.append(float(l64.detach()))
  out['loss64'] = np.array(loss64)
  out['dproj64'] = np.array([recipe.projection_signs(seed, i, P[k].numel(), K_PROJ).astype(np.float64) @
                             (P[k].detach() - p0d[k]).reshape(-1).numpy() for i, k in enumerate(names)])
  for k, v in P.items():
    if k.endswith('running_mean') or k.endswith('running_var'):
      out['bn64/' + k] = v.numpy().copy()
  print('float64 trajectory: losses', ['%.8f' % v for v in loss64])
  nbt = {}
  for k, v in net.state_dict().items():
    if k.endswith('running_mean') or k.endswith('running_var'):
      out['bn/' + k] = v.numpy().copy()
    elif k.endswith('num_batches_tracked'):
      nbt[k] = int(v)
  # the shared extractor runs twice per step (left, right: mode_disparity.py:100-101): 2 K batches there, K in the 3-D stage
  assert set(nbt.values()) == {K_STEPS, 2 * K_STEPS} and all((v == 2 * K_STEPS) == k.startswith('feature_extraction') for k, v in nbt.items())
  out['nbt'] = np.array([K_STEPS, 2 * K_STEPS])
  path = os.path.join(HERE, 'model_steps_tiny.npz')
  np.savez_compressed(path, **out)
  print('wrote %s (%.0f KB)' % (path, os.path.getsize(path) / 1024))


if __name__ == '__main__':
  main()
